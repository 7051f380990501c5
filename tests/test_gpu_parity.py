"""GPU parity: the HIP path through the C ABI against the CPU oracle, bit for bit.

MSM results are group elements, so they are compared after affine normalisation (Jacobian
representatives are not unique); NTT output is unique and compared bytewise
(SURVEY.md section 8(b)).  BN254 is the curve the reference proves over (pinned oracle);
Pallas/Vesta curve results are "parity unpinned" in the reference -- the oracle they are
compared with is pinned only through its field arithmetic (tests/test_oracle_pins.py).
"""
import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu

CURVES = ["bn254", "pallas", "vesta"]
CID = O.CURVE_IDS
SEED = 0x48324D5300000000


def scalar_field(curve):
    return R.CURVES[curve].scalar


def omega_limbs(curve, log_n, inverse=False):
    f = scalar_field(curve)
    w = f.omega(log_n)
    if inverse:
        w = pow(w, -1, f.p)
    return np.array(f.limbs(w), dtype=np.uint64)


def rand_scalars(curve, n, seed=1):
    return O.synth_scalars(O.CURVE_SCALAR_FIELD[CID[curve]], SEED | seed, n).reshape(n, 4)


def rand_bases(curve, n, seed=0xB5):
    return O.synth_bases(CID[curve], SEED | seed, n).reshape(n, 8)


def norm(curve, jac):
    return O.to_affine(CID[curve], np.asarray(jac, dtype=np.uint64).reshape(-1))


# ------------------------------------------------------------------------------- NTT ----
@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("log_n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17])
def test_ntt_matches_oracle(h2, curve, log_n):
    n = 1 << log_n
    a = rand_scalars(curve, n, seed=log_n)
    w = omega_limbs(curve, log_n)
    want = O.best_fft(O.CURVE_SCALAR_FIELD[CID[curve]], a, w, log_n, threads=4).reshape(n, 4)
    got = a.copy()
    h2.best_fft(got, w, log_n, curve)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("curve", ["bn254", "pallas"])
def test_ntt_large_two_and_three_pass(h2, curve):
    for log_n in (18, 20, 21):
        n = 1 << log_n
        a = rand_scalars(curve, n, seed=log_n)
        w = omega_limbs(curve, log_n)
        want = O.best_fft(O.CURVE_SCALAR_FIELD[CID[curve]], a, w, log_n, threads=8).reshape(n, 4)
        got = a.copy()
        h2.best_fft(got, w, log_n, curve)
        assert np.array_equal(got, want), log_n


def test_ntt_inverse_round_trip_and_batch(h2):
    curve, log_n = "bn254", 12
    n = 1 << log_n
    f = scalar_field(curve)
    cols = [rand_scalars(curve, n, seed=10 + j) for j in range(3)]
    orig = [c.copy() for c in cols]
    h2.best_fft_batch(cols, omega_limbs(curve, log_n), log_n, curve)
    for c, o in zip(cols, orig):
        want = O.best_fft(1, o, omega_limbs(curve, log_n), log_n).reshape(n, 4)
        assert np.array_equal(c, want)
    h2.best_fft_batch(cols, omega_limbs(curve, log_n, inverse=True), log_n, curve)
    # iNTT(NTT(a)) = n * a  (best_fft does not scale; EvaluationDomain::ifft multiplies by n^-1)
    n_m = np.array(f.limbs(n), dtype=np.uint64)
    for c, o in zip(cols, orig):
        want = O.field_mul_many(1, o.reshape(-1), np.tile(n_m, n)).reshape(n, 4)
        assert np.array_equal(c, want)


@pytest.mark.parametrize("curve", ["bn254", "pallas"])
def test_scaled_ntt_all_plan_shapes(h2, curve):
    """h2_ntt_scaled_device (EvaluationDomain::ifft's 1/n and other constants): one pass multiplies in its final pass,
    two passes take the constant from inter-pass twiddles built with it (a table per (omega, log n, constant)), three
    passes multiply in the final pass again.  Two constants against the same omega must not share a table; the
    unscaled transform of the same omega in between must stay unscaled."""
    import ctypes
    import torch
    f = scalar_field(curve)
    fid = O.CURVE_SCALAR_FIELD[CID[curve]]
    L = h2.load()
    for log_n in (3, 9, 10, 11, 13, 16, 20, 21):
        n = 1 << log_n
        a = rand_scalars(curve, n, seed=0x30 + log_n)
        w = omega_limbs(curve, log_n, inverse=True)
        plain = O.best_fft(fid, a, w, log_n, threads=8).reshape(n, 4)
        for c in (pow(n, -1, f.p), 0x123456789ABCDEF0FEDCBA9876543210 % f.p, 1):
            cm = np.array(f.limbs(c), dtype=np.uint64)
            d = torch.from_numpy(a.view(np.int64)).cuda()
            st = L.h2_ntt_scaled_device(CID[curve], ctypes.c_void_p(d.data_ptr()), 1, w.ctypes.data, log_n, cm.ctypes.data, None)
            assert st == 0
            torch.cuda.synchronize()
            want = O.field_mul_many(fid, plain.reshape(-1), np.tile(cm, n)).reshape(n, 4)
            assert np.array_equal(d.cpu().numpy().view(np.uint64), want), (log_n, hex(c))
            d2 = torch.from_numpy(a.view(np.int64)).cuda()
            h2.ntt_device(d2.data_ptr(), 1, w, log_n, curve)
            torch.cuda.synchronize()
            assert np.array_equal(d2.cpu().numpy().view(np.uint64), plain), log_n


# ---------------------------------------------------------------------- group-element FFT ----
def _affine_to_jac(curve, aff):
    """(n, 8) affine points -> (n, 12) Jacobian with z = 1 (Montgomery), identity (0, 0) -> z = 0"""
    f = R.CURVES[curve].base
    n = aff.shape[0]
    jac = np.zeros((n, 12), dtype=np.uint64)
    jac[:, :8] = aff
    one = np.array(f.limbs(1), dtype=np.uint64)
    for i in range(n):
        if aff[i].any():
            jac[i, 8:] = one
    return jac


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8])
def test_group_fft_matches_oracle(h2, curve, log_n):
    """h2_fft_group == best_fft over C::Curve (the FftGroup impl behind g_to_lagrange), against the oracle's restatement;
    inputs include the identity, a repeated point and a point next to its negative (exceptional additions)"""
    n = 1 << log_n
    c = R.CURVES[curve]
    aff = rand_bases(curve, n, seed=0xC0 + log_n)
    if n >= 8:
        aff[2] = 0
        aff[5] = aff[4]
        neg = aff[6].copy()
        y = c.base.from_mont(O.limbs_to_int(neg[4:]))
        neg[4:] = np.array(c.base.limbs((-y) % c.base.p), dtype=np.uint64)
        aff[7] = neg
    jac = _affine_to_jac(curve, aff)
    w = omega_limbs(curve, log_n, inverse=True)
    want = O.to_affine(CID[curve], O.group_fft(CID[curve], jac.reshape(-1), w, log_n)).reshape(n, 8)
    got = jac.copy()
    h2.best_fft_group(got, w, log_n, curve)
    assert np.array_equal(O.to_affine(CID[curve], got.reshape(-1)).reshape(n, 8), want)


@pytest.mark.parametrize("k", [4, 6])
def test_group_fft_reproduces_g_lagrange_of_the_recorded_params(h2, k):
    """ParamsKZG::new's g_to_lagrange on the reference's own params files (sha256-pinned): g_lagrange = n^-1 *
    best_fft(g, omega^-1, k).  The transform's output is compared with [n] g_lagrange, every point."""
    import os
    f = R.BN_FR
    n = 1 << k
    data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "params_k%d.bin" % k), "rb").read()
    g = np.frombuffer(data, dtype=np.uint64, count=8 * n, offset=4).reshape(n, 8).copy()
    gl = np.frombuffer(data, dtype=np.uint64, count=8 * n, offset=4 + 64 * n).reshape(n, 8).copy()
    jac = _affine_to_jac("bn254", g)
    h2.best_fft_group(jac, omega_limbs("bn254", k, inverse=True), k, "bn254")
    got = O.to_affine(0, jac.reshape(-1)).reshape(n, 8)
    n_m = np.array(f.limbs(n), dtype=np.uint64)
    for i in range(n):
        assert np.array_equal(got[i], O.to_affine(0, O.scalar_mul(0, n_m, gl[i]))), i
    with pytest.raises(ValueError):
        h2.best_fft_group(jac, omega_limbs("bn254", k), k + 1, "bn254")


def test_ntt_rejects_bad_length(h2):
    a = rand_scalars("bn254", 8)
    with pytest.raises(ValueError):
        h2.best_fft(a, omega_limbs("bn254", 4), 4, "bn254")


# ------------------------------------------------------------------------------- MSM ----
@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("n", [1, 2, 3, 5, 31, 33, 1000, 2048, 1 << 14])
def test_msm_matches_oracle(h2, curve, n):
    s = rand_scalars(curve, n, seed=n & 0xFF)
    b = rand_bases(curve, n)
    want = norm(curve, O.best_multiexp(CID[curve], s, b, threads=8))
    got = norm(curve, h2.best_multiexp(s, b, curve))
    assert np.array_equal(got, want)
    assert O.is_on_curve(CID[curve], got)


def test_msm_2_16_bn254_and_pallas(h2):
    for curve in ("bn254", "pallas"):
        n = 1 << 16
        s = rand_scalars(curve, n, seed=7)
        b = rand_bases(curve, n)
        want = norm(curve, O.best_multiexp(CID[curve], s, b, threads=8))
        got = norm(curve, h2.best_multiexp(s, b, curve))
        assert np.array_equal(got, want)


@pytest.mark.parametrize("curve", ["bn254", "pallas"])
def test_msm_edge_scalars(h2, curve):
    """zeros are skipped, ones / p-1 / small values hit the sign and carry paths."""
    n = 300
    f = scalar_field(curve)
    b = rand_bases(curve, n)
    vals = [0, 1, 2, f.p - 1, f.p - 2, (1 << 128) - 1, 1 << 253, (f.p - 1) // 2, (f.p + 1) // 2, 0x8000, 0x7FFF,
            0xFFFF, 0x10000]
    s = np.array([f.limbs(vals[i % len(vals)]) for i in range(n)], dtype=np.uint64)
    want = norm(curve, O.best_multiexp(CID[curve], s, b))
    got = norm(curve, h2.best_multiexp(s, b, curve))
    assert np.array_equal(got, want)
    # all-zero column -> identity (0,0), as fixed_commitments print `Infinity`
    z = np.zeros((n, 4), dtype=np.uint64)
    assert not norm(curve, h2.best_multiexp(z, b, curve)).any()


@pytest.mark.parametrize("curve", ["bn254", "pallas"])
def test_msm_group_law_collisions(h2, curve):
    """duplicate bases (P + P in one bucket), P and -P with equal scalars (P - P), identity bases."""
    n = 64
    c = R.CURVES[curve]
    b = rand_bases(curve, n)
    b[1] = b[0]                       # same point twice
    b[3] = b[2]
    neg = b[4].copy()
    y = c.base.from_mont(O.limbs_to_int(neg[4:]))
    neg[4:] = np.array(c.base.limbs((-y) % c.base.p), dtype=np.uint64)
    b[5] = neg                        # -P next to P
    b[6] = 0                          # identity base
    s = rand_scalars(curve, n, seed=3)
    s[1] = s[0]
    s[3] = s[2]
    s[5] = s[4]
    want = norm(curve, O.best_multiexp(CID[curve], s, b))
    got = norm(curve, h2.best_multiexp(s, b, curve))
    assert np.array_equal(got, want)
    # everything cancels: s*P + s*(-P) = identity
    b2 = np.stack([b[4], neg])
    s2 = np.stack([s[4], s[4]])
    assert not norm(curve, h2.best_multiexp(s2, b2, curve)).any()


def test_msm_hot_bucket_all_ones(h2):
    """degenerate witness column: every scalar equal -> one bucket per window holds all n points."""
    curve, n = "bn254", 1 << 13
    f = scalar_field(curve)
    b = rand_bases(curve, n)
    for v in (1, 0x1234567):
        s = np.tile(np.array(f.limbs(v), dtype=np.uint64), (n, 1))
        want = norm(curve, O.best_multiexp(CID[curve], s, b, threads=8))
        got = norm(curve, h2.best_multiexp(s, b, curve))
        assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [1 << 16, 1 << 18])
def test_msm_degenerate_columns_take_the_hierarchical_path(h2, n):
    """a permutation grand product that is 1 on almost every row, a 0/1 selector, two repeated values mixed with
    dense rows: buckets with tens of thousands of entries (more than MSM_HOT_SPAN chunk pieces)"""
    curve = "bn254"
    f = scalar_field(curve)
    b = rand_bases(curve, n)
    bases = h2.Bases(curve, b)
    try:
        one = np.array(f.limbs(1), dtype=np.uint64)
        ones = np.tile(one, (n, 1))
        ones[-6:] = rand_scalars(curve, 6, seed=77)                 # blinding rows
        sel = np.zeros((n, 4), dtype=np.uint64)
        sel[::3] = one                                             # 0/1 selector
        two_vals = np.tile(np.array(f.limbs(0x1234567 << 40), dtype=np.uint64), (n, 1))
        two_vals[1::2] = np.array(f.limbs(f.p - 5), dtype=np.uint64)
        mixed = rand_scalars(curve, n, seed=78)
        mixed[: n // 2] = one
        cols = [ones, sel, two_vals, mixed]
        got = bases.msm_batch(cols)
        for j, col in enumerate(cols):
            want = norm(curve, O.best_multiexp(CID[curve], col, b, threads=8))
            assert np.array_equal(got[j], want), j
        # one column alone (different chunk size T than the batch)
        assert np.array_equal(norm(curve, bases.msm(ones)), norm(curve, O.best_multiexp(CID[curve], ones, b, threads=8)))
    finally:
        bases.release()


def test_msm_sparse_witness_shape(h2):
    """zero except 64 dense rows at the top and 6 at the bottom (SURVEY.md section 8(d) 'sparse')."""
    curve, n = "bn254", 1 << 12
    b = rand_bases(curve, n)
    s = np.zeros((n, 4), dtype=np.uint64)
    d = rand_scalars(curve, 70, seed=9)
    s[:64] = d[:64]
    s[-6:] = d[64:]
    want = norm(curve, O.best_multiexp(CID[curve], s, b))
    got = norm(curve, h2.best_multiexp(s, b, curve))
    assert np.array_equal(got, want)


def test_msm_resident_bases_prefix_and_batch(h2):
    curve, n = "bn254", 4096
    b = rand_bases(curve, n)
    bases = h2.Bases(curve, b)
    try:
        assert bases.plan()["windows"] * bases.plan()["window_bits"] >= 255
        cols = [rand_scalars(curve, n, seed=20 + j) for j in range(5)]
        cols[3][:] = 0
        got = bases.msm_batch(cols)
        for j, col in enumerate(cols):
            want = norm(curve, O.best_multiexp(CID[curve], col, b, threads=8))
            assert np.array_equal(got[j], want), j
        # a shorter column uses a prefix of the registered bases
        short = cols[0][:1000]
        want = norm(curve, O.best_multiexp(CID[curve], short, b[:1000], threads=4))
        assert np.array_equal(norm(curve, bases.msm(short)), want)
        with pytest.raises(ValueError):
            bases.msm(rand_scalars(curve, n + 1))
    finally:
        bases.release()


def test_msm_many_columns_take_the_multi_kernel_scan(h2):
    """70 columns of 2^13 in ONE launch: 70 x 512 buckets = 35 840 keys, more than the one-block LDS scan holds, so the
    per-XCD counters go through msm_group_fold_kernel and the three scan kernels; 512 buckets per column also give the
    row / column weights an uneven split (32 rows x 16 columns)."""
    curve, n, m = "pallas", 1 << 13, 70
    b = rand_bases(curve, n)
    bases = h2.Bases(curve, b)
    try:
        assert bases.plan()["window_bits"] == 10
        cols = [rand_scalars(curve, n, seed=300 + j) for j in range(m)]
        cols[7][:] = 0
        cols[8][1:] = 0
        got = bases.msm_batch(cols)
        for j in (0, 1, 7, 8, 33, 68, 69):
            want = norm(curve, O.best_multiexp(CID[curve], cols[j], b, threads=8))
            assert np.array_equal(got[j], want), j
    finally:
        bases.release()


def test_msm_length_mismatch_is_an_error(h2):
    with pytest.raises(ValueError):
        h2.best_multiexp(rand_scalars("bn254", 4), rand_bases("bn254", 5), "bn254")
    import ctypes
    lib = h2.load()
    out = np.zeros(12, dtype=np.uint64)
    s = rand_scalars("bn254", 4)
    assert lib.h2_msm(0, 0xDEAD, s.ctypes.data, 4, out.ctypes.data) == -4   # H2_EHANDLE
    assert lib.h2_ntt(7, s.ctypes.data, s.ctypes.data, 2) == -1             # H2_EINVAL
    assert isinstance(ctypes.c_char_p(lib.h2_strerror(-1)).value, bytes)


# ------------------------------------------------------------- device field arithmetic ----
@pytest.mark.parametrize("name", list(R.FIELDS))
def test_device_field_ops_match_oracle(h2, name):
    """the gfx950 Comba multiplier (and add/sub/neg/inv) on edge values and random elements"""
    import random
    f = R.FIELDS[name]
    fid = O.FIELD_IDS[name]
    rng = random.Random(fid + 99)
    special = [0, 1, 2, f.p - 1, f.p - 2, (1 << 255) % f.p, (1 << 32) - 1, 1 << 32, (1 << 64) - 1, 1 << 224,
               ((1 << 256) - 1) % f.p, f.p >> 1, (f.p >> 1) + 1, (1 << 253) - 1]
    vals = special + [rng.randrange(f.p) for _ in range(4096 - len(special))]
    n = len(vals)
    a = np.array([f.limbs(v) for v in vals], dtype=np.uint64)
    b = np.array([f.limbs(vals[(7 * i + 3) % n]) for i in range(n)], dtype=np.uint64)
    # every special value against every special value
    sa = np.array([f.limbs(x) for x in special for _ in special], dtype=np.uint64)
    sb = np.array([f.limbs(y) for _ in special for y in special], dtype=np.uint64)
    a, b = np.concatenate([a, sa]), np.concatenate([b, sb])
    n = a.shape[0]
    lib = h2.load()
    out = np.zeros_like(a)
    for op, oname in ((2, "mul"), (0, "add"), (1, "sub"), (7, "mul"), (9, "mul")):   # 7: CIOS, 9: 29-bit working form
        assert lib.h2_selftest_field_op_device(fid, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n) == 0
        if oname == "mul":
            want = O.field_mul_many(fid, a.reshape(-1), b.reshape(-1)).reshape(n, 4)
        else:
            want = np.array([O.field_op(fid, oname, a[i], b[i]) for i in range(n)], dtype=np.uint64)
        assert np.array_equal(out, want), oname
    k = 64
    assert lib.h2_selftest_field_op_device(fid, 3, a[1:k + 1].copy().ctypes.data, b.ctypes.data, out.ctypes.data, k) == 0
    for i in range(k):
        assert O.limbs_to_int(out[i]) == f.to_mont(pow(vals[i + 1], -1, f.p))


@pytest.mark.parametrize("curve", ["bn254", "pallas", "vesta"])
def test_device_group_law_on_the_working_form(h2, curve):
    """The MSM's working representation (9 x 29-bit limbs, csrc/h2_curve29.hpp) on the device, against big integers:
    the 4-lanes-per-point addition / doubling (ops 0, 1, 4), the one-lane forms (2, 3) and the weight kernel's
    double-and-add with a different multiplier per quad (5) -- including P + P, P - P and identity operands."""
    c = R.CURVES[curve]
    cid = O.CURVE_IDS[curve]
    L = h2.load()
    n = 28
    Ps = [c.mul(0x1234567 + 3 * i, c.gen) for i in range(n)]
    Qs = [c.mul(0x7654321 + 5 * i, c.gen) for i in range(n)]
    Qs[3] = Ps[3]                  # P + P inside an addition
    Qs[4] = c.neg(Ps[4])           # P - P
    Qs[5] = None                   # P + O
    Ps[6] = None                   # O + Q
    Ps[7] = Qs[7] = None           # O + O

    def aff(pts):
        return np.frombuffer(b"".join(c.affine_bytes(P) for P in pts), dtype=np.uint64).reshape(len(pts), 8).copy()

    p, q = aff(Ps), aff(Qs)
    out = np.zeros_like(p)
    dbl = [c.add(a, a) for a in Ps]
    for op, want in ((0, [c.add(a, b) for a, b in zip(Ps, Qs)]), (1, dbl), (2, [c.add(a, b) for a, b in zip(Ps, Qs)]),
                     (3, dbl), (4, [c.add(d, b) for d, b in zip(dbl, Qs)])):
        assert L.h2_selftest_curve_op_device(cid, op, p.ctypes.data, q.ctypes.data, out.ctypes.data, n) == 0
        assert np.array_equal(out, aff(want)), op
    ks = [0, 1, 2, 3, 4, 5, 6, 7, 8, 31, 32, 33, 255, 256, 257, 4095, 4096, 4097, 0xFFFF, 0x10001, 0xABCDE, 2047, 2048,
          1023, 77, 0xFFFFFFFF, 0x80000000, 0xAAAAAAAA]
    kq = np.zeros((n, 8), dtype=np.uint64)
    kq[:, 0] = ks
    kq[:, 4] = 1
    assert L.h2_selftest_curve_op_device(cid, 5, p.ctypes.data, kq.ctypes.data, out.ctypes.data, n) == 0
    assert np.array_equal(out, aff([c.mul(k, P) if k else None for k, P in zip(ks, Ps)]))
