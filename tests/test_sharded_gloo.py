"""N > 1 path on CPU: world_size-2 gloo ranks shard the columns of a commit phase, all-gather the
commitments, and every rank must hold the same vector as a single-rank run (oracle as the MSM backend:
the product has no CPU path, the sharding / gather logic is what is under test)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m, n, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_lib as O
    from halo2_prover_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bases = O.synth_bases(0, 0x48324D53000000B5, n).reshape(n, 8)
    cols = [O.synth_scalars(1, 0x48324D5300000100 + j, n).reshape(n, 4) for j in range(m)]
    cols[1][:] = 0   # an all-zero column commits to the identity

    def oracle_batch(cs):
        return np.stack([O.to_affine(0, O.best_multiexp(0, c, bases)) for c in cs]) if cs else np.zeros((0, 8), np.uint64)

    got = sharded.commit_columns(None, cols, msm_batch=oracle_batch)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), got)
    # a single MSM split by contiguous point range: partial sums add up to the whole
    lo, hi = sharded.split_msm_by_range(n, rank, world)
    part = O.best_multiexp(0, cols[0][lo:hi], bases[lo:hi])
    np.save(os.path.join(out_dir, "part%d.npy" % rank), part)
    dist.barrier()
    dist.destroy_process_group()


class _HostBases:
    """stand-in for api.Bases over HOST memory with the oracle as the arithmetic: what is under test is the sharding
    logic of sharded.msm_phase_device (ranges, strides, gather order, partial-sum addition), which is the same code
    that drives h2_msm_device_range / h2_points_sum_device on the GPUs"""

    def __init__(self, O, bases):
        self.O, self.bases = O, bases

    @staticmethod
    def _view(ptr, count):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint64 * count).from_address(ptr))

    def msm_device_range(self, d_scalars, first, n, stride, m, d_out, stream=0):
        out = self._view(d_out, 12 * m).reshape(m, 12)
        for j in range(m):
            col = self._view(d_scalars + j * stride * 32, 4 * n).reshape(n, 4)
            out[j] = self.O.best_multiexp(0, col.copy(), self.bases[first:first + n])

    def points_sum_device(self, d_in, groups, count, d_out, stream=0):
        src = self._view(d_in, 12 * groups * count).reshape(groups, count, 12)
        out = self._view(d_out, 12 * count).reshape(count, 12)
        for j in range(count):
            acc = np.zeros(12, dtype=np.uint64)
            for g in range(groups):
                acc = self.O.jac_add(0, acc, src[g, j].copy())
            out[j] = acc


def _phase_worker(rank, world, port, m, n, mode, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    from halo2_prover_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bases = O.synth_bases(0, 0x48324D53000000B5, n).reshape(n, 8)
    cols = np.stack([O.synth_scalars(1, 0x48324D5300000200 + j, n).reshape(n, 4) for j in range(m)])
    if m > 1:
        cols[1][:] = 0
    t = torch.from_numpy(cols.view(np.int64).copy())
    got = sharded.msm_phase_device(_HostBases(O, bases), t.data_ptr(), n, m, mode=mode, device="cpu")
    aff = np.stack([O.to_affine(0, r) for r in got.numpy().view(np.uint64)])
    np.save(os.path.join(out_dir, "phase%d.npy" % rank), aff)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,m,mode", [(2, 1, None), (2, 4, None), (2, 5, None), (3, 2, None), (3, 3, None),
                                          (2, 4, "range"), (3, 5, "columns")])
def test_phase_sharding_by_columns_and_by_point_range(tmp_path, world, m, mode):
    """every rank ends with the same m commitments as an unsharded run, whichever way the phase was split"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    from halo2_prover_amd import sharded
    n = 50                               # not a multiple of the world sizes: ragged point ranges
    mp.spawn(_phase_worker, args=(world, _free_port(), m, n, mode, str(tmp_path)), nprocs=world, join=True)
    bases = O.synth_bases(0, 0x48324D53000000B5, n).reshape(n, 8)
    cols = [O.synth_scalars(1, 0x48324D5300000200 + j, n).reshape(n, 4) for j in range(m)]
    if m > 1:
        cols[1][:] = 0
    want = np.stack([O.to_affine(0, O.best_multiexp(0, c, bases)) for c in cols])
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "phase%d.npy" % r)), want), r
    assert sharded.phase_mode(4, 2) == "columns" and sharded.phase_mode(5, 2) == "range"
    assert sharded.phase_mode(1, 8) == "range" and sharded.phase_mode(3, 1) == "single"


@pytest.mark.parametrize("world,m", [(2, 5), (2, 4), (3, 2)])
def test_sharded_commit_matches_single_rank(tmp_path, world, m):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    n = 64
    port = _free_port()
    mp.spawn(_worker, args=(world, port, m, n, str(tmp_path)), nprocs=world, join=True)
    bases = O.synth_bases(0, 0x48324D53000000B5, n).reshape(n, 8)
    cols = [O.synth_scalars(1, 0x48324D5300000100 + j, n).reshape(n, 4) for j in range(m)]
    cols[1][:] = 0
    want = np.stack([O.to_affine(0, O.best_multiexp(0, c, bases)) for c in cols])
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, want), r
    assert not want[1].any()
    total = np.zeros(12, dtype=np.uint64)
    for r in range(world):
        total = O.jac_add(0, total, np.load(os.path.join(str(tmp_path), "part%d.npy" % r)))
    assert np.array_equal(O.to_affine(0, total), want[0])


def test_shard_assignment_covers_every_column_once():
    from halo2_prover_amd import sharded
    for world in (1, 2, 3, 8):
        for m in (0, 1, 4, 5, 16, 64):
            seen = sorted(j for r in range(world) for j in sharded.shard_columns(m, r, world))
            assert seen == list(range(m))
    for world in (1, 2, 4, 8):
        n = 1 << 20
        edges = [sharded.split_msm_by_range(n, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
