"""GPU: BASELINE.json's full sizes through size-independent properties (the oracle is too slow there), and the
golden params fixtures through ParamsKZG.

* MSM n = 2^20 (config 4): linearity  MSM(a) + MSM(b) = MSM(a + b)  and  MSM(c, c, ..., c) = c * sum(P_i)
  with the sum obtained from an MSM of ones; bases from h2_srs_generate (itself checked against the oracle at
  n = 2^8, whose [s^i]G is pinned by the params sha256).
* MSM n = 2^24 (config 5's column length; dense and "realistic witness" sparse columns): known answer
  MSM(a, [s^i]G) = (sum_i a_i s^i) G with the scalar from the oracle's eval_polynomial.
* column groups: a batch whose sort would not fit 32-bit entry indices is run in groups of columns
  (the h2_selftest_set_msm_max_entries hook forces that path at a small size) and must equal the ungrouped result.
* config 5's per-GPU shape itself: 8 columns of 2^24 through the column groups, and NTT 2^24 x 2 columns.
* NTT n = 2^22 (three passes) and the 64-column batch shape of config 5 at reduced n: iNTT(NTT(a)) = n a,
  linearity, and A[0] = sum(a).
"""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def L(f, v):
    return np.array(f.limbs(v), dtype=np.uint64)


def srs(h2, curve, s, n):
    import torch
    f = R.CURVES[curve].scalar
    buf = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    s_m = L(f, s)
    st = h2.load().h2_srs_generate(h2.CURVES[curve], s_m.ctypes.data, n, ctypes.c_void_p(buf.data_ptr()), None)
    assert st == 0
    torch.cuda.synchronize()
    return buf


def test_srs_generate_matches_pinned_powers(h2):
    n = 256
    f = R.BN_FR
    s = R.SurveyStream().fr_random(f)      # the scalar behind the pinned params files
    got = srs(h2, "bn254", s, n).cpu().numpy().view(np.uint64)
    want = O.to_affine(0, O.powers_of_s(0, L(f, s), n)).reshape(n, 8)
    assert np.array_equal(got, want)
    data = open(os.path.join(GOLDEN, "params_k6.bin"), "rb").read()
    g = np.frombuffer(data, dtype=np.uint64, count=8 * 64, offset=4).reshape(64, 8)
    assert np.array_equal(got[:64], g)     # == the g vector of the reference's own params file (sha256-pinned)


def test_params_kzg_on_golden_files(h2):
    """ParamsKZG.read/write round trip and commit / commit_lagrange on the pinned k = 4, 6 params."""
    f = R.BN_FR
    for k in (4, 6):
        data = open(os.path.join(GOLDEN, "params_k%d.bin" % k), "rb").read()
        params = h2.ParamsKZG.read(data)
        assert params.k == k and params.write() == data
        n = 1 << k
        ones = np.tile(L(f, 1), (n, 1))
        # sum_i g_lagrange[i] = g[0]  
        assert np.array_equal(O.to_affine(0, params.commit_lagrange(ones)), params.g[0])
        # commit(e_i) = g[i]; commit_lagrange(w^(-ij)/n) = g_lagrange row relation
        for i in (0, 1, n - 1):
            e = np.zeros((n, 4), dtype=np.uint64)
            e[i] = L(f, 1)
            assert np.array_equal(O.to_affine(0, params.commit(e)), params.g[i])
        winv, ninv = pow(f.omega(k), -1, f.p), pow(n, -1, f.p)
        i = 3
        coeffs = np.array([f.limbs(pow(winv, i * j, f.p) * ninv % f.p) for j in range(n)], dtype=np.uint64)
        assert np.array_equal(O.to_affine(0, params.commit(coeffs)), params.g_lagrange[i])
        # a batch of columns in one launch, against the oracle
        cols = [O.synth_scalars(1, 0x48324D5300000300 + c, n).reshape(n, 4) for c in range(4)]
        got = params.commit_many(cols, lagrange=True)
        for c, col in enumerate(cols):
            assert np.array_equal(got[c], O.to_affine(0, O.best_multiexp(0, col, params.g_lagrange)))
        with pytest.raises(ValueError):
            params.commit(ones[:-1])
    with pytest.raises(ValueError):
        h2.ParamsKZG.read(data[:-1])


@pytest.mark.parametrize("curve", ["pallas", "bn254"])
def test_msm_2_20_properties(h2, curve):
    import torch
    cid = O.CURVE_IDS[curve]
    fs = R.CURVES[curve].scalar
    n = 1 << 20
    bases = h2.Bases.from_device(curve, srs(h2, curve, 0xABCDEF0123, n).data_ptr(), n)
    try:
        assert bases.plan()["window_bits"] == 16
        a = O.synth_scalars(O.CURVE_SCALAR_FIELD[cid], 0x48324D5300000401, n).reshape(n, 4)
        b = O.synth_scalars(O.CURVE_SCALAR_FIELD[cid], 0x48324D5300000402, n).reshape(n, 4)
        dev = torch.from_numpy(np.stack([a, b]).view(np.int64)).cuda()
        # a + b on the device (pointwise add through the ABI), so the sum column never leaves HBM
        s_dev = dev[0].clone()
        st = h2.load().h2_poly_pointwise_device(cid, 0, ctypes.c_void_p(s_dev.data_ptr()), ctypes.c_void_p(dev[1].data_ptr()),
                                                n, None)
        assert st == 0
        cols = torch.stack([dev[0], dev[1], s_dev]).contiguous()
        out = torch.zeros((3, 12), dtype=torch.int64, device="cuda")
        bases.msm_device(cols.data_ptr(), n, 3, out.data_ptr())
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint64)
        pa, pb, pab = res[0], res[1], res[2]
        assert np.array_equal(O.to_affine(cid, O.jac_add(cid, pa, pb)), O.to_affine(cid, pab))
        assert O.is_on_curve(cid, O.to_affine(cid, pab)) and O.to_affine(cid, pab).any()
        # constant column: MSM(c, ..., c) = c * MSM(1, ..., 1)
        c = 0x1D2C3B4A5F6E7788990011223344556677
        ones = np.tile(L(fs, 1), (n, 1))
        consts = np.tile(L(fs, c), (n, 1))
        p1 = O.to_affine(cid, bases.msm(ones))
        pc = O.to_affine(cid, bases.msm(consts))
        assert np.array_equal(O.to_affine(cid, O.scalar_mul(cid, L(fs, c), p1)), pc)
    finally:
        bases.release()


def test_msm_2_24_known_answer(h2):
    import torch
    curve = "pallas"
    cid = O.CURVE_IDS[curve]
    fid = O.CURVE_SCALAR_FIELD[cid]
    fs = R.CURVES[curve].scalar
    n = 1 << 24
    s = 0x2B7E151628AED2A6ABF7158809CF4F3C762E7160F38B4DA56A784D9045190CFE % fs.p
    g = srs(h2, curve, s, n)
    gen = g[0].cpu().numpy().view(np.uint64)                    # [s^0]G = the generator
    bases = h2.Bases.from_device(curve, g.data_ptr(), n)
    try:
        dense = O.synth_scalars(fid, 0x48324D5300000500, n).reshape(n, 4)
        sparse = np.zeros((n, 4), dtype=np.uint64)              # 64 dense rows on top, 6 at the bottom
        sparse[:64] = dense[:64]
        sparse[n - 6:] = dense[n - 6:]
        cols = torch.from_numpy(np.stack([dense, sparse]).view(np.int64)).cuda()
        out = torch.zeros((2, 12), dtype=torch.int64, device="cuda")
        bases.msm_device(cols.data_ptr(), n, 2, out.data_ptr())
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint64)
        for j, col in enumerate((dense, sparse)):
            k = O.eval_polynomial(fid, col, L(fs, s))
            want = O.to_affine(cid, O.scalar_mul(cid, k, gen))
            assert np.array_equal(O.to_affine(cid, res[j]), want), "column %d" % j
    finally:
        bases.release()


def test_msm_column_groups_equal_one_launch(h2):
    import torch
    curve = "bn254"
    cid = O.CURVE_IDS[curve]
    fid = O.CURVE_SCALAR_FIELD[cid]
    n, m = 1 << 10, 5
    b = O.synth_bases(cid, 0x48324D53000006B5, n).reshape(n, 8)
    bases = h2.Bases(curve, b)
    try:
        cols = np.stack([O.synth_scalars(fid, 0x48324D5300000600 + j, n).reshape(n, 4) for j in range(m)])
        whole = bases.msm_batch(list(cols))
        windows = bases.plan()["windows"]
        L = h2.load()
        L.h2_selftest_set_msm_max_entries(2 * windows * n)                  # two columns per launch: groups 2 + 2 + 1
        grouped = bases.msm_batch(list(cols))
        dev = torch.from_numpy(cols.view(np.int64)).cuda()
        out = torch.zeros((m, 12), dtype=torch.int64, device="cuda")
        bases.msm_device(dev.data_ptr(), n, m, out.data_ptr())
        torch.cuda.synchronize()
        L.h2_selftest_set_msm_max_entries(windows * n - 1)                  # not even one column fits: rejected
        with pytest.raises(h2.H2Error) as err:
            bases.msm_batch(list(cols))
        assert err.value.status == -1                                       # H2_EINVAL
        L.h2_selftest_set_msm_max_entries(0)
        assert np.array_equal(whole, grouped)
        jac = out.cpu().numpy().view(np.uint64)
        for j in range(m):
            assert np.array_equal(O.to_affine(cid, jac[j]), whole[j])
            assert np.array_equal(whole[j], O.to_affine(cid, O.best_multiexp(cid, cols[j], b)))
    finally:
        bases.release()


def test_ntt_2_22_and_batch_properties(h2):
    import torch
    curve = "pallas"
    cid = O.CURVE_IDS[curve]
    f = R.CURVES[curve].scalar
    fid = O.CURVE_SCALAR_FIELD[cid]
    for log_n, m in ((22, 1), (16, 64)):
        n = 1 << log_n
        a = O.synth_scalars(fid, 0x48324D5300000500 + log_n, n * m).reshape(m, n, 4)
        w, winv = L(f, f.omega(log_n)), L(f, pow(f.omega(log_n), -1, f.p))
        d = torch.from_numpy(a.view(np.int64)).cuda()
        h2.ntt_device(d.data_ptr(), m, w, log_n, curve)
        torch.cuda.synchronize()
        fwd = d.cpu().numpy().view(np.uint64).reshape(m, n, 4)
        # A[0] = sum_j a[j]  (column 0 and the last column)
        for col in (0, m - 1):
            # exact sum of the column by pairwise halving with the device's pointwise add
            s = torch.from_numpy(a[col].view(np.int64)).cuda()
            length = n
            while length > 1:
                half = length // 2
                lo, hi = s[:half].contiguous(), s[half:length].contiguous()
                assert h2.load().h2_poly_pointwise_device(cid, 0, ctypes.c_void_p(lo.data_ptr()),
                                                          ctypes.c_void_p(hi.data_ptr()), half, None) == 0
                s, length = lo, half
            torch.cuda.synchronize()
            assert np.array_equal(s[0].cpu().numpy().view(np.uint64), fwd[col][0])
        # round trip: iNTT(NTT(a)) = n * a
        h2.ntt_device(d.data_ptr(), m, winv, log_n, curve)
        torch.cuda.synchronize()
        back = d.cpu().numpy().view(np.uint64).reshape(m, n, 4)
        sample = np.random.RandomState(1).randint(0, n, size=64)
        n_m = L(f, n)
        for col in (0, m - 1):
            want = O.field_mul_many(fid, a[col][sample].reshape(-1), np.tile(n_m, 64)).reshape(64, 4)
            assert np.array_equal(back[col][sample], want)
        # one column of the batch against the oracle outright (n = 2^16), and spot columns equal single-column runs
        if log_n == 16:
            want = O.best_fft(fid, a[5], w, log_n, threads=8).reshape(n, 4)
            assert np.array_equal(fwd[5], want)


def test_config5_per_gpu_shape_2_24(h2):
    """BASELINE.json config 5, one GPU's share: 8 columns of 2^24 rows.  The MSM batch is 1.88e9 sort entries (known answer: every column is the same dense column, so all eight commitments equal
    (sum_i a_i s^i) G); the NTT of 2 columns of 2^24 (three passes) is checked through A[0] = sum(a) on a sparse
    column, linearity against a second column, and the round trip."""
    import torch
    curve = "pallas"
    cid = O.CURVE_IDS[curve]
    fid = O.CURVE_SCALAR_FIELD[cid]
    fs = R.CURVES[curve].scalar
    n, m = 1 << 24, 8
    s = 0x1F83D9ABFB41BD6B5BE0CD19137E2179A54FF53A5F1D36F1510E527FADE682D1 % fs.p
    g = srs(h2, curve, s, n)
    gen = g[0].cpu().numpy().view(np.uint64)
    bases = h2.Bases.from_device(curve, g.data_ptr(), n)
    try:
        # 14 windows of 19 bits: 8 * 14 * 2^24 = 1.88e9 sort entries, the largest single launch the library makes (the
        # limit is 2^31 - 1; with the 16 windows of round 2 this batch ran as two column groups -- that path is
        # test_msm_column_groups_equal_one_launch's)
        assert (1 << 30) < m * bases.plan()["windows"] * n < (1 << 31)
        dense = O.synth_scalars(fid, 0x48324D5300000501, n).reshape(n, 4)
        d1 = torch.from_numpy(dense.view(np.int64)).cuda()
        cols = d1.unsqueeze(0).expand(m, n, 4).contiguous()
        out = torch.zeros((m, 12), dtype=torch.int64, device="cuda")
        bases.msm_device(cols.data_ptr(), n, m, out.data_ptr())
        torch.cuda.synchronize()
        del cols
        res = out.cpu().numpy().view(np.uint64)
        want = O.to_affine(cid, O.scalar_mul(cid, O.eval_polynomial(fid, dense, L(fs, s)), gen))
        for j in range(m):
            assert np.array_equal(O.to_affine(cid, res[j]), want), j
    finally:
        bases.release()
    del g
    torch.cuda.empty_cache()
    # NTT 2^24 x 2 columns: column 0 sparse (64 + 6 non-zero rows), column 1 dense
    log_n = 24
    a = np.zeros((2, n, 4), dtype=np.uint64)
    a[0, :64] = dense[:64]
    a[0, n - 6:] = dense[n - 6:]
    a[1] = dense
    w, winv = L(fs, fs.omega(log_n)), L(fs, pow(fs.omega(log_n), -1, fs.p))
    d = torch.from_numpy(a.view(np.int64)).cuda()
    h2.ntt_device(d.data_ptr(), 2, w, log_n, curve)
    torch.cuda.synchronize()
    rows = np.array([0, 1, 12345, n // 2, n - 1])
    fwd = d[:, torch.from_numpy(rows).cuda()].cpu().numpy().view(np.uint64).reshape(2, len(rows), 4)
    nz = [i for i in list(range(64)) + list(range(n - 6, n))]
    vals = [fs.from_mont(O.limbs_to_int(dense[i])) for i in nz]
    wv = fs.omega(log_n)
    for r, k in enumerate(rows):                                   # A[k] = sum_j a[j] w^(jk), 70 terms
        want = sum(v * pow(wv, (j * int(k)) % n, fs.p) for v, j in zip(vals, nz)) % fs.p
        assert fs.from_mont(O.limbs_to_int(fwd[0][r])) == want, k
    h2.ntt_device(d.data_ptr(), 2, winv, log_n, curve)
    torch.cuda.synchronize()
    sample = np.random.RandomState(2).randint(0, n, size=64)
    back = d[:, torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint64).reshape(2, 64, 4)
    n_m = L(fs, n)
    for col in (0, 1):
        want = O.field_mul_many(fid, a[col][sample].reshape(-1), np.tile(n_m, 64)).reshape(64, 4)
        assert np.array_equal(back[col], want)
