"""CPU: the MSM launch geometry, checked on the host for a sweep of shapes (no GPU).

Round 2 recorded a GPU memory-access fault under an uncommitted build of the XCD-grouped sort.
Since then every launch runs `msm_check` first: each kernel's largest index against the bytes of the region it indexes,
grid sizes, dynamic LDS.  These tests run that proof, and the sort's block -> (column, tile) mapping, for the shapes the
library can meet: odd column counts, ragged lengths, prefixes of the registered bases, both sorts, both scatter kernels.
The device side of the same shapes (red zones behind every region) is tests/test_gpu_msm_geometry.py.
"""
import ctypes

import numpy as np
import pytest

import halo2_prover_amd as h2


@pytest.fixture(scope="module")
def lib():
    return h2.load()


def check(lib, curve, n_bases, n, m, stride=None, guard=0):
    out = (ctypes.c_uint64 * 8)()
    st = lib.h2_selftest_msm_check(curve, n_bases, n, m, n if stride is None else stride, guard, out)
    return st, dict(zip(("c", "W", "B", "tile", "staged", "sort2", "T", "regions"), [int(x) for x in out]))


def test_tile_mapping_is_a_bijection_and_surplus_blocks_are_dead(lib):
    """the grid is rounded up to a multiple of 8 blocks; a surplus block that took a (column, tile) pair would read
    scalars past the last column -- the fault's most likely cause"""
    for tiles in list(range(1, 41)) + [63, 64, 65, 127, 255, 256, 257, 1000, 4097]:
        for m in list(range(1, 10)) + [16, 17, 64, 70]:
            if tiles * m <= (1 << 24):
                assert lib.h2_selftest_msm_tiles(tiles, m) == 0, (tiles, m)


@pytest.mark.parametrize("curve", [0, 1, 2])
def test_every_shape_passes_the_bounds_proof(lib, curve):
    shapes = 0
    seen = set()
    for k in range(1, 25):
        for n_bases in {1 << k, (1 << k) - 1, (1 << k) + 1, 3 * (1 << k) // 2 + 1}:
            if n_bases < 1:
                continue
            for n in {1, max(1, n_bases // 3), max(1, n_bases - 5), n_bases}:
                for m in (1, 2, 3, 5, 7, 16, 70):
                    st, info = check(lib, curve, n_bases, n, m)
                    W = info["W"]
                    if W and (W * n * m >= (1 << 31) or info["B"] * m >= (1 << 31)):
                        continue          # msm_device_run splits such batches into column groups before the layout
                    assert st == 0, (n_bases, n, m, lib.h2_last_device_error())
                    st, _ = check(lib, curve, n_bases, n, m, guard=1)
                    assert st == 0, (n_bases, n, m, "guard", lib.h2_last_device_error())
                    seen.add((info["staged"], info["sort2"]))
                    shapes += 1
    assert shapes > 2000
    assert seen == {(0, 0), (1, 0), (0, 1)}      # direct scatter, staged scatter, two-level sort all met


def test_the_three_sorts_are_chosen_where_designed(lib):
    _, a = check(lib, 1, 1 << 16, 1 << 16, 4)
    assert (a["c"], a["B"], a["staged"], a["sort2"]) == (12, 2048, 1, 0)
    _, b = check(lib, 1, 1 << 20, 1 << 20, 1)
    assert (b["c"], b["W"], b["B"], b["sort2"]) == (16, 16, 32768, 1)
    _, d = check(lib, 1, 1 << 22, 1 << 22, 3)
    assert (d["c"], d["W"], d["B"], d["sort2"]) == (19, 14, 1 << 18, 1)
    _, c = check(lib, 1, 3000, 3000, 2)
    assert (c["staged"], c["sort2"]) == (0, 0)


def test_bad_strides_and_lengths_are_refused(lib):
    assert check(lib, 0, 1 << 12, 1 << 12, 3, stride=(1 << 12) - 1)[0] == -1      # columns would overlap
    assert b"col_stride" in lib.h2_last_device_error()
    assert check(lib, 0, 1 << 12, (1 << 12) + 1, 1)[0] == -1                        # longer than the bases
    assert check(lib, 0, 1 << 12, 0, 1)[0] == -1
    assert check(lib, 0, 1 << 12, 16, 0)[0] == -1
    assert check(lib, 9, 1 << 12, 16, 1)[0] == -1
