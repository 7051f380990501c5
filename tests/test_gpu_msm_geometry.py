"""GPU: MSM launches of awkward shapes with a red zone behind every region of the scratch arena.

The shapes are those the round-2 review asked for after the recorded memory-access fault:
tiles * m not a multiple of 8 (surplus blocks of the XCD-grouped grid), m = 3 / 5 / 7 at 2^16, ragged lengths, prefixes
of the registered bases, the staged and the direct scatter, the two-level sort, first call with the profiling events
on.  `h2_selftest_msm_guard(1)` lays the arena out with 256 guard bytes behind every region, fills it with a pattern
before each launch sequence and counts the guard bytes that changed; results are compared with the CPU oracle.
"""
import ctypes

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SEED = 0x48324D5300000000
CID = O.CURVE_IDS


def scalars(curve, n, seed):
    return O.synth_scalars(O.CURVE_SCALAR_FIELD[CID[curve]], SEED | seed, n).reshape(n, 4)


def bases_of(curve, n, seed=0xB5):
    return O.synth_bases(CID[curve], SEED | seed, n, threads=8).reshape(n, 8)


def guard_report(lib):
    out = (ctypes.c_uint64 * 2)()
    first = ctypes.create_string_buffer(256)
    assert lib.h2_selftest_msm_guard_report(out, first, 256) == 0
    return int(out[0]), int(out[1]), first.value.decode()


@pytest.fixture()
def guarded(h2):
    lib = h2.load()
    lib.h2_selftest_msm_guard(1)
    yield lib
    lib.h2_selftest_msm_guard(0)


# (curve, registered bases, column length, columns, columns checked against the oracle)
SHAPES = [
    ("bn254", 1 << 16, 1 << 16, 3, (0, 2)),            # staged scatter; tiles * m odd multiples
    ("pallas", 1 << 16, 1 << 16, 5, (4,)),
    ("bn254", 1 << 16, 1 << 16, 7, (6,)),
    ("pallas", 1 << 16, (1 << 16) - 5, 3, (1,)),       # ragged: a prefix of the bases, last tile short
    ("bn254", 1 << 16, 40001, 1, (0,)),
    ("bn254", 3000, 3000, 3, (0, 1, 2)),                # direct scatter (too few scalars for the staged one)
    ("pallas", 3000, 2999, 5, (3,)),
    ("bn254", 1 << 12, 1 << 12, 5, (0, 4)),
    ("vesta", 1 << 13, 1 << 13, 9, (8,)),
    ("bn254", 1 << 18, 1 << 18, 1, (0,)),               # two-level sort (8192 buckets)
    ("pallas", 1 << 18, (1 << 18) - 77, 3, (2,)),
    ("bn254", 300001, 299999, 2, (1,)),
    ("bn254", 1 << 21, (1 << 21) - 3, 2, (1,)),         # 19-bit windows: 2^18 buckets = 1024 coarse bins x 256 fine
    ("pallas", 1 << 18, 1 << 18, 9, (0, 7, 8)),         # 9 columns x 1024 coarse bins > 8192: two launches (8 + 1) of the two-level sort
]


@pytest.mark.parametrize("curve,n_bases,n,m,verify", SHAPES)
def test_awkward_shapes_stay_inside_their_regions(h2, guarded, curve, n_bases, n, m, verify):
    lib = guarded
    lib.h2_profile_enable(1)                            # the roofline events ride along, as in bench.py's timed region
    b = bases_of(curve, n_bases)
    bases = h2.Bases(curve, b)
    try:
        cols = [scalars(curve, n, 40 + j) for j in range(m)]
        if m > 1:
            cols[1][n // 2:] = 0                        # a half-empty column: fewer entries than the worst case
        got = bases.msm_batch(cols)
        launches, violations, first = guard_report(lib)
        assert launches >= 1
        assert violations == 0, first
        for j in verify:
            want = O.to_affine(CID[curve], O.best_multiexp(CID[curve], cols[j], b[:n], threads=8))
            assert np.array_equal(got[j], want), j
    finally:
        lib.h2_profile_enable(0)
        bases.release()


@pytest.mark.parametrize("n_bases,first,n", [(1 << 18, 1 << 17, 1 << 17), (1 << 18, 12345, 200000), (1 << 16, 30000, 35536)])
def test_point_ranges_of_the_bases(h2, guarded, n_bases, first, n):
    """h2_msm_device_range: bases [first, first + n) -- one rank's share of a range-split MSM.  The sorted entries are
    w * n_bases + i with the REGISTERED length (a two-level sort that sized its packed entries for the range's length
    let the key bits overlap them: found by the 2-rank bench at 2^20)"""
    import torch
    lib = guarded
    curve = "bn254"
    b = bases_of(curve, n_bases)
    bases = h2.Bases(curve, b)
    try:
        cols = np.stack([scalars(curve, n_bases, 70 + j) for j in range(2)])
        d = torch.from_numpy(cols.view(np.int64)).cuda()
        out = torch.zeros((2, 12), dtype=torch.int64, device="cuda")
        bases.msm_device_range(d.data_ptr() + first * 32, first, n, n_bases, 2, out.data_ptr())
        torch.cuda.synchronize()
        launches, violations, msg = guard_report(lib)
        assert launches >= 1 and violations == 0, msg
        res = out.cpu().numpy().view(np.uint64)
        for j in range(2):
            want = O.to_affine(CID[curve], O.best_multiexp(CID[curve], cols[j][first:first + n].copy(), b[first:first + n].copy(), threads=8))
            assert np.array_equal(O.to_affine(CID[curve], res[j]), want), j
    finally:
        bases.release()


def test_two_level_sort_with_the_side_array(h2, guarded):
    """at 2^24 bases an entry (w * n + i | sign) has no spare bits for the key's low bits, which then travel in a byte
    array; guard(3) selects that layout at 2^18"""
    lib = guarded
    lib.h2_selftest_msm_guard(3)
    curve, n = "pallas", 1 << 18
    b = bases_of(curve, n)
    bases = h2.Bases(curve, b)
    try:
        cols = [scalars(curve, n, 60 + j) for j in range(2)]
        cols[1][::2] = 0
        got = bases.msm_batch(cols)
        launches, violations, first = guard_report(lib)
        assert launches >= 1 and violations == 0, first
        for j, col in enumerate(cols):
            assert np.array_equal(got[j], O.to_affine(CID[curve], O.best_multiexp(CID[curve], col, b, threads=8))), j
    finally:
        bases.release()


def test_two_level_sort_on_skewed_columns(h2, guarded):
    """coarse bins far from uniform: a constant column (every entry of a window in ONE bucket: bins of 2^18 entries, many
    slabs per level-2 block), a 0/1 selector, a sparse column, and a dense one, 2^18 rows each"""
    import pyref as R
    lib = guarded
    curve, n = "bn254", 1 << 18
    f = R.CURVES[curve].scalar
    b = bases_of(curve, n)
    bases = h2.Bases(curve, b)
    try:
        const = np.tile(np.array(f.limbs(0x1234567 << 40), dtype=np.uint64), (n, 1))
        sel = np.zeros((n, 4), dtype=np.uint64)
        sel[::3] = np.array(f.limbs(1), dtype=np.uint64)
        sparse = np.zeros((n, 4), dtype=np.uint64)
        sparse[:64] = scalars(curve, 64, 91)
        sparse[-6:] = scalars(curve, 6, 92)
        dense = scalars(curve, n, 93)
        cols = [const, sel, sparse, dense]
        got = bases.msm_batch(cols)
        launches, violations, first = guard_report(lib)
        assert launches >= 1 and violations == 0, first
        for j, col in enumerate(cols):
            want = O.to_affine(CID[curve], O.best_multiexp(CID[curve], col, b, threads=8))
            assert np.array_equal(got[j], want), j
        one = bases.msm(const)                          # alone: another chunk size, another bin population
        assert np.array_equal(O.to_affine(CID[curve], one), got[0])
        assert guard_report(lib)[1] == 0
    finally:
        bases.release()


def test_guard_mode_does_catch_an_overrun(h2, guarded):
    """guard(2) makes the library itself write one byte behind the second region: the checker must report it"""
    lib = guarded
    curve, n = "bn254", 1 << 10
    b = bases_of(curve, n)
    bases = h2.Bases(curve, b)
    try:
        bases.msm(scalars(curve, n, 5))
        assert guard_report(lib) == (1, 0, "")
        lib.h2_selftest_msm_guard(2)
        bases.msm(scalars(curve, n, 5))
        launches, violations, first = guard_report(lib)
        assert (launches, violations) == (1, 1) and "1 byte(s) behind the region" in first, first
    finally:
        bases.release()


def test_launch_sequences_of_changing_shape_share_one_workspace(h2):
    """without the guard mode's fill: a launch sequence leaves its counter region zero for the next one on the same
    workspace, which skips its memset when its own region is no larger (Arena::clean_bytes).  Shapes whose layouts
    differ -- more columns, fewer, another sort, another bucket count -- follow each other here; every result is checked"""
    curve = "bn254"
    nb = {12: bases_of(curve, 1 << 12, 0xC1), 13: bases_of(curve, 1 << 13, 0xC2), 18: bases_of(curve, 1 << 18, 0xC3)}
    regs = {k: h2.Bases(curve, v) for k, v in nb.items()}
    try:
        plan = [(12, 1 << 12, 5), (12, 1 << 12, 2), (13, 1 << 13, 3), (12, 1 << 12, 5), (12, 3000, 1), (18, 1 << 18, 2),
                (12, 1 << 12, 4), (13, (1 << 13) - 9, 7), (12, 1 << 12, 4), (18, 200000, 1), (12, 100, 3), (12, 1 << 12, 5)]
        for step, (k, n, m) in enumerate(plan):
            cols = [scalars(curve, n, 0x300 + 16 * step + j) for j in range(m)]
            if m > 2:
                cols[2][: n // 3] = 0
            got = regs[k].msm_batch(cols)
            for j in (0, m - 1):
                want = O.to_affine(CID[curve], O.best_multiexp(CID[curve], cols[j], nb[k][:n], threads=8))
                assert np.array_equal(got[j], want), (step, k, n, m, j)
    finally:
        for r in regs.values():
            r.release()
