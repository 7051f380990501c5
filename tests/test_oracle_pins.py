"""Pins of the CPU oracle (no GPU): the reference's own vectors and recorded outputs.

* Pasta Fp/Fq arithmetic: the 44 zcash Poseidon vectors and the constants-table spot values the
  reference's tests hold (tests/golden/pasta_poseidon_vectors.json, extracted by
  tests/golden/make_pasta_poseidon_vectors.py).
* BN254 Fr arithmetic + Grain/MDS: values recorded from the reference's build in SURVEY.md App. B.5.
* BN254 Fq, G1 group law, NTT root/order/scaling, wire format: sha256 of the params file recorded in
  SURVEY.md App. B.2 for k = 4, 6, 10 (11, 16 under -m slow) re-derived by the C oracle; best_multiexp
  itself against the same params (g_lagrange[i] = MSM(row i of the inverse DFT matrix, g)).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import pyref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAMS_SHA256 = {
    4: "e410bf985e9327e7ea474d74907e4209b50678844fab11bc09bcd1e2a2ae1272",
    6: "3cd009bb91fe7f1d4c2cc54296263062e5a1d2359aacd68bfa1ce542b5a1169d",
    10: "24cef0fa77991930622fce4c51c7ddf40aaf3324779b1c6592c1c29e6043374b",
    11: "c071f033c580c8d827fb719c4d428a0d10673b4ab7da0c2ea62dec3ffc3fc6ca",
    16: "07d2055cadf19515cc5e2bdc14a46d54b8fccb37e5da5afa2a58cccb0012cee8",
}


def le(hexstr):
    return int.from_bytes(bytes.fromhex(hexstr), "little")


@pytest.mark.parametrize("name,field", [("fp", R.PA_FP), ("fq", R.PA_FQ)])
def test_pasta_poseidon_vectors(name, field):
    v = json.load(open(os.path.join(GOLDEN, "pasta_poseidon_vectors.json")))[name]
    rcs, mds = R.poseidon_constants(field, 3, 8, 56)
    assert "%064x" % rcs[0][0] == v["round_constant_0_0"]
    assert "%064x" % mds[0][0] == v["mds_0_0"]
    assert len(v["permute"]) == 11 and len(v["hash"]) == 11
    for t in v["permute"]:
        out = R.poseidon_permute(field, [le(x) for x in t["initial_state"]], rcs, mds, 8, 56)
        assert out == [le(x) for x in t["final_state"]]
    for t in v["hash"]:
        assert R.poseidon_hash_const_len(field, [le(x) for x in t["input"]], rcs, mds, 8, 56) == le(t["output"])


def test_pasta_poseidon_through_the_c_oracle_field_ops():
    """the same permutation evaluated with the C oracle's Montgomery mul/add (second vector of each field)."""
    v = json.load(open(os.path.join(GOLDEN, "pasta_poseidon_vectors.json")))
    for name, field in (("fp", R.PA_FP), ("fq", R.PA_FQ)):
        fid = O.FIELD_IDS["pasta_" + name]
        rcs, mds = R.poseidon_constants(field, 3, 8, 56)
        L = lambda x: np.array(field.limbs(x), dtype=np.uint64)  # noqa: E731
        mul = lambda a, b: O.field_op(fid, "mul", a, b)          # noqa: E731
        add = lambda a, b: O.field_op(fid, "add", a, b)          # noqa: E731
        t = v[name]["permute"][1]
        st = [L(le(x)) for x in t["initial_state"]]
        for r in range(64):
            full = r < 4 or r >= 60
            st = [add(st[i], L(rcs[r][i])) for i in range(3)]
            for i in range(3 if full else 1):
                x2 = mul(st[i], st[i])
                st[i] = mul(mul(x2, x2), st[i])
            st = [add(add(mul(L(mds[i][0]), st[0]), mul(L(mds[i][1]), st[1])), mul(L(mds[i][2]), st[2]))
                  for i in range(3)]
        assert [field.from_mont(O.limbs_to_int(s)) for s in st] == [le(x) for x in t["final_state"]]


def test_bn254_recorded_values():
    f = R.BN_FR
    rcs, mds = R.poseidon_constants(f, 3, 8, 60)
    assert rcs[0][0] == 0x0EE1F344726EBD994115140900A1A52518B73D0D534B720A6CE9B336BDC8F841
    assert mds[0][0] == 0x108C4512F46F539B7732B2A44D80530AE468104492496B33E4162D1C84956E45
    assert R.poseidon_hash_const_len(f, [1, 2], rcs, mds, 8, 60) == \
        0x152E960B5C9C8A624B2CDF4855250E8A54EE074254281310DC4A9704F78C1917
    assert f.omega(4) == 0x21082CA216CBBF4E1C6E4F4594DD508C996DFBE1174EFB98B11509C6E306460B
    assert f.root_of_unity == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C


def params_bytes_c(k):
    """ParamsKZG::new(k).write() re-derived with the C oracle under the SURVEY App. B.2 RNG stream."""
    f = R.BN_FR
    s = R.SurveyStream().fr_random(f)
    n = 1 << k
    L = lambda x: np.array(f.limbs(x), dtype=np.uint64)  # noqa: E731
    g = O.powers_of_s(0, L(s), n)
    gl = O.group_fft(0, g, L(pow(f.omega(k), -1, f.p)), k)
    gl = O.scale_points(0, L(pow(n, -1, f.p)), gl)
    ga, gla = O.to_affine(0, g), O.to_affine(0, gl)
    tail = R.g2_bytes(R.G2_GEN) + R.g2_bytes(R.g2_mul(s, R.G2_GEN))
    return k.to_bytes(4, "little") + ga.tobytes() + gla.tobytes() + tail


@pytest.mark.parametrize("k", [4, 6, 10])
def test_params_sha256_matches_reference_record(k):
    data = params_bytes_c(k)
    assert len(data) == 4 + 128 * (1 << k) + 256
    assert hashlib.sha256(data).hexdigest() == PARAMS_SHA256[k]


def test_golden_params_files_are_the_pinned_bytes():
    for k in (4, 6):
        data = open(os.path.join(GOLDEN, "params_k%d.bin" % k), "rb").read()
        assert hashlib.sha256(data).hexdigest() == PARAMS_SHA256[k]


def test_pyref_agrees_on_params_k4():
    s = R.SurveyStream().fr_random(R.BN_FR)
    data_py, _, _ = R.params_kzg_bytes(4, s)
    assert hashlib.sha256(data_py).hexdigest() == PARAMS_SHA256[4]


def test_best_multiexp_reproduces_g_lagrange():
    """g_lagrange[i] = sum_j (w^(-ij)/n) g[j]: every output of the pinned params file is an MSM KAT."""
    for k in (4, 6):
        data = open(os.path.join(GOLDEN, "params_k%d.bin" % k), "rb").read()
        n = 1 << k
        g = np.frombuffer(data, dtype=np.uint64, count=8 * n, offset=4).reshape(n, 8)
        gl = np.frombuffer(data, dtype=np.uint64, count=8 * n, offset=4 + 64 * n).reshape(n, 8)
        f = R.BN_FR
        winv, ninv = pow(f.omega(k), -1, f.p), pow(n, -1, f.p)
        for i in (0, 1, n // 2, n - 1):
            coeffs = np.array([f.limbs(pow(winv, i * j, f.p) * ninv % f.p) for j in range(n)], dtype=np.uint64)
            for threads in (1, 3):
                got = O.to_affine(0, O.best_multiexp(0, coeffs, g, threads=threads))
                assert np.array_equal(got, gl[i]), (k, i, threads)
        # sum_i g_lagrange[i] = g[0]
        ones = np.tile(np.array(f.limbs(1), dtype=np.uint64), (n, 1))
        assert np.array_equal(O.to_affine(0, O.best_multiexp(0, ones, gl)), g[0])


def test_best_fft_matches_naive_dft_and_pyref():
    for name, fid in O.FIELD_IDS.items():
        f = R.FIELDS[name]
        if f.S < 6:
            continue
        for log_n in (1, 3, 6):
            n = 1 << log_n
            rng = R.SplitMix64(99 + log_n)
            a = [R.synth_scalar(rng, f.p) for _ in range(n)]
            w = f.omega(log_n)
            want = R.dft_naive(a, w, f.p)
            assert R.best_fft(a, w, log_n, f.p) == want
            arr = np.array([f.limbs(x) for x in a], dtype=np.uint64)
            for threads in (1, 4):
                got = O.best_fft(fid, arr, np.array(f.limbs(w), dtype=np.uint64), log_n, threads=threads)
                assert [f.from_mont(O.limbs_to_int(got[4 * i:4 * i + 4])) for i in range(n)] == want


def test_synthetic_generators_agree_between_pyref_and_c():
    for name, fid in O.FIELD_IDS.items():
        f = R.FIELDS[name]
        rng = R.SplitMix64(0x48324D5300000001)
        want = [R.synth_scalar(rng, f.p) for _ in range(8)]
        got = O.synth_scalars(fid, 0x48324D5300000001, 8)
        assert [f.from_mont(O.limbs_to_int(got[4 * i:4 * i + 4])) for i in range(8)] == want


def test_eval_polynomial_matches_big_int_horner():
    for name, fid in O.FIELD_IDS.items():
        f = R.FIELDS[name]
        rng = R.SplitMix64(77)
        co = [R.synth_scalar(rng, f.p) for _ in range(33)]
        x = R.synth_scalar(rng, f.p)
        want = sum(c * pow(x, i, f.p) for i, c in enumerate(co)) % f.p
        got = O.eval_polynomial(fid, np.array([f.limbs(c) for c in co], dtype=np.uint64), np.array(f.limbs(x), dtype=np.uint64))
        assert f.from_mont(O.limbs_to_int(got)) == want
        assert not O.eval_polynomial(fid, np.zeros((0, 4), dtype=np.uint64), np.array(f.limbs(x), dtype=np.uint64)).any()


def test_oracle_msm_pasta_self_consistency():
    """Pallas/Vesta curve results are 'parity unpinned' in the reference: self-consistency only."""
    for name in ("pallas", "vesta"):
        c = R.CURVES[name]
        cid = O.CURVE_IDS[name]
        n = 40
        rng = R.SplitMix64(5)
        scal = [R.synth_scalar(rng, c.scalar.p) for _ in range(n)]
        pts = [c.mul(3 + 5 * i, c.gen) for i in range(n)]
        want = c.msm(scal, pts)
        s = np.array([c.scalar.limbs(x) for x in scal], dtype=np.uint64)
        b = np.frombuffer(b"".join(c.affine_bytes(P) for P in pts), dtype=np.uint64).reshape(n, 8)
        got = O.to_affine(cid, O.best_multiexp(cid, s, b, threads=2))
        assert got.tobytes() == c.affine_bytes(want)
        sb = O.synth_bases(cid, 1234, 16).reshape(16, 8)
        assert O.is_on_curve(cid, sb)


@pytest.mark.slow
def test_params_sha256_k11_and_k16():
    for k in (11, 16):
        assert hashlib.sha256(params_bytes_c(k)).hexdigest() == PARAMS_SHA256[k]
