"""GPU: the multi-GPU forms of the hot path, rehearsed on the box's one GPU.

* a commit phase split over two ranks (one process per rank, sharing cuda:0, gloo): by whole columns and by point
  range (BASELINE config 4's split of one MSM) -- every rank must end with the unsplit result, bit for bit after
  normalisation;
* h2_msm_device_range / h2_points_sum_device against the oracle;
* one process, two contexts (h2_init_devices(2, {0, 0})): h2_msm_batch sharded by column, h2_msm split by range,
  h2_ntt_batch sharded by column -- the C ABI's own multi-GPU mode (include/h2hip.h);
* two caller streams at once (the arenas' event hand-over): MSMs and NTTs issued on two streams give the same bytes
  as one stream;
* `bench.py --gpus 2` starts its own ranks and reports n_gpus = 2.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x48324D5300000000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _omega(curve, log_n):
    f = R.CURVES[curve].scalar
    return np.array(f.limbs(f.omega(log_n)), dtype=np.uint64)


def _phase_worker(rank, world, port, curve, n, m, mode, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import halo2_prover_amd as h2
    from halo2_prover_amd import sharded
    h2.init(0)
    cid = O.CURVE_IDS[curve]
    b = O.synth_bases(cid, SEED | 0x7B5, n).reshape(n, 8)
    cols = np.stack([O.synth_scalars(O.CURVE_SCALAR_FIELD[cid], SEED | (0x700 + j), n).reshape(n, 4) for j in range(m)])
    bases = h2.Bases(curve, b)
    d = torch.from_numpy(cols.view(np.int64)).cuda()
    # the stream contract: everything is ordered on torch's current stream -- here a non-default one for the sharded
    # call (looked up when no handle is given) and the default stream for the unsplit one; a foreign handle is refused
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = sharded.msm_phase_device(bases, d.data_ptr(), n, m, mode=mode)
        try:
            sharded.msm_phase_device(bases, d.data_ptr(), n, m, 0, mode=mode)
            raise AssertionError("a stream other than the current one was accepted")
        except ValueError:
            pass
    torch.cuda.current_stream().wait_stream(side)
    whole = sharded.msm_phase_device(bases, d.data_ptr(), n, m, 0, mode="single")
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "got%d.npy" % rank), got.cpu().numpy().view(np.uint64))
    np.save(os.path.join(out_dir, "whole%d.npy" % rank), whole.cpu().numpy().view(np.uint64))
    bases.release()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("curve,n,m,mode", [("bn254", 1 << 14, 1, "range"), ("pallas", 1 << 14, 3, "range"),
                                            ("bn254", 1 << 18, 1, "range"),      # the two-level sort on a point range
                                            ("bn254", 5000, 4, "columns"), ("bn254", 5000, 5, None)])
def test_phase_split_over_two_ranks_equals_the_unsplit_result(tmp_path, curve, n, m, mode):
    world = 2
    mp.spawn(_phase_worker, args=(world, _free_port(), curve, n, m, mode, str(tmp_path)), nprocs=world, join=True)
    cid = O.CURVE_IDS[curve]
    b = O.synth_bases(cid, SEED | 0x7B5, n).reshape(n, 8)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "got%d.npy" % r))
        whole = np.load(os.path.join(str(tmp_path), "whole%d.npy" % r))
        for j in range(m):
            assert np.array_equal(O.to_affine(cid, got[j]), O.to_affine(cid, whole[j])), (r, j)
    col0 = O.synth_scalars(O.CURVE_SCALAR_FIELD[cid], SEED | 0x700, n).reshape(n, 4)
    assert np.array_equal(O.to_affine(cid, got[0]), O.to_affine(cid, O.best_multiexp(cid, col0, b, threads=4)))


def test_msm_device_range_and_points_sum_match_the_oracle(h2):
    import torch
    curve, n, m = "bn254", 3000, 3
    cid = O.CURVE_IDS[curve]
    b = O.synth_bases(cid, SEED | 0x8B5, n).reshape(n, 8)
    cols = np.stack([O.synth_scalars(1, SEED | (0x800 + j), n).reshape(n, 4) for j in range(m)])
    bases = h2.Bases(curve, b)
    try:
        d = torch.from_numpy(cols.view(np.int64)).cuda()
        parts = torch.zeros((3, m, 12), dtype=torch.int64, device="cuda")
        edges = [(0, 1), (1, 1777), (1777, n)]                  # ragged ranges, the first a single point
        for g, (lo, hi) in enumerate(edges):
            bases.msm_device_range(d.data_ptr() + lo * 32, lo, hi - lo, n, m, parts[g].data_ptr())
        out = torch.zeros((m, 12), dtype=torch.int64, device="cuda")
        bases.points_sum_device(parts.data_ptr(), 3, m, out.data_ptr())
        torch.cuda.synchronize()
        p = parts.cpu().numpy().view(np.uint64)
        for g, (lo, hi) in enumerate(edges):
            for j in range(m):
                want = O.best_multiexp(cid, cols[j][lo:hi].copy(), b[lo:hi].copy())
                assert np.array_equal(O.to_affine(cid, p[g, j]), O.to_affine(cid, want)), (g, j)
        res = out.cpu().numpy().view(np.uint64)
        for j in range(m):
            assert np.array_equal(O.to_affine(cid, res[j]), O.to_affine(cid, O.best_multiexp(cid, cols[j], b)))
        # a range that leaves the registered vector is rejected
        with pytest.raises(h2.H2Error) as err:
            bases.msm_device_range(d.data_ptr(), n - 10, 11, n, 1, out.data_ptr())
        assert err.value.status == -1
    finally:
        bases.release()


_TWO_CONTEXTS = r'''
import os, sys
import numpy as np
ROOT = %r
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle_lib as O
import pyref as R
import halo2_prover_amd as h2
from halo2_prover_amd import api
api.init_devices([0, 0])                      # two contexts on the one GPU
assert h2.load().h2_device_count() == 2
SEED = 0x48324D5300000000
for curve in ("bn254", "pallas"):
    cid = O.CURVE_IDS[curve]
    fid = O.CURVE_SCALAR_FIELD[cid]
    n, m = 1 << 14, 5
    b = O.synth_bases(cid, SEED | 0x9B5, n).reshape(n, 8)
    cols = [O.synth_scalars(fid, SEED | (0x900 + j), n).reshape(n, 4) for j in range(m)]
    bases = h2.Bases(curve, b)
    got = bases.msm_batch(cols)               # columns 0, 2, 4 on context 0; 1, 3 on context 1
    for j in range(m):
        assert np.array_equal(got[j], O.to_affine(cid, O.best_multiexp(cid, cols[j], b, threads=4))), (curve, j)
    one = bases.msm(cols[0])                  # one MSM: split by point range over the two contexts
    assert np.array_equal(O.to_affine(cid, one), got[0])
    short = bases.msm(cols[1][:100])          # too short to split: context 0 alone
    assert np.array_equal(O.to_affine(cid, short), O.to_affine(cid, O.best_multiexp(cid, cols[1][:100].copy(), b[:100].copy())))
    f = R.CURVES[curve].scalar
    w = np.array(f.limbs(f.omega(14)), dtype=np.uint64)
    tcols = [c.copy() for c in cols[:3]]
    h2.best_fft_batch(tcols, w, 14, curve)
    for c, o in zip(tcols, cols[:3]):
        assert np.array_equal(c, O.best_fft(fid, o, w, 14, threads=4).reshape(n, 4))
    bases.release()
print("two contexts ok")
'''


def test_one_process_two_contexts_shard_the_host_pointer_entry_points():
    """h2_init_devices in a fresh process (the context list is per process and the other tests use h2_init(0))"""
    r = subprocess.run([sys.executable, "-c", _TWO_CONTEXTS % ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "two contexts ok" in r.stdout, r.stdout + r.stderr


_SHARDED_PROVER = r'''
import ctypes, hashlib, os, sys
ROOT = %r
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import halo2_prover_amd as h2
from halo2_prover_amd import api
import test_capi_product as T
api.init_devices([0, 0])                      # two contexts on the one GPU
L = h2.load()
assert L.h2_device_count() == 2
L.h2_selftest_set_shard_min_rows(8)           # spread even the k = 4 proof's commitments (16 rows) over the contexts
golden = {4: T.golden("params_k4.bin"), 6: T.golden("params_k6.bin")}
done = 0
for name, k, js, idx in (("arithmetic", 4, T.ARITH_INPUT, 1), ("poseidon", 6, T.POSEIDON_INPUT, 2),
                         ("collatz", 10, T.COLLATZ_INPUT, 0), ("poseidon", 11, T.POSEIDON_INPUT, 2),
                         ("poseidon", 16, T.POSEIDON_INPUT, 2)):
    rng = T.Stream(0)
    params = T.c_setup(L, k, rng)
    assert hashlib.sha256(params).hexdigest() == T.PARAMS_SHA256[k], k
    if k in golden:
        assert params == golden[k]
    before = L.h2_selftest_sharded_commits()
    proof = T.c_prove(L, params, js, idx, rng)
    assert L.h2_selftest_sharded_commits() > before, "the commit phases were not spread over the two contexts"
    assert hashlib.sha256(proof).hexdigest() == T.PROOF_SHA256[(name, k)], (name, k)
    assert T.c_verify(L, params, proof, js, idx) == (0, 1)
    # the default threshold: small proofs stay on one context, k >= 11 is spread; the bytes do not change
    L.h2_selftest_set_shard_min_rows(0)
    before = L.h2_selftest_sharded_commits()
    rng = T.Stream(0)
    T.c_setup(L, k, rng)
    assert hashlib.sha256(T.c_prove(L, params, js, idx, rng)).hexdigest() == T.PROOF_SHA256[(name, k)]
    assert (L.h2_selftest_sharded_commits() > before) == (k >= 11), k
    L.h2_selftest_set_shard_min_rows(8)
    done += 1
print("sharded prover ok", done)
'''


def test_cpp_prover_spreads_its_commit_phases_over_two_contexts_and_keeps_the_bytes():
    """h2_generate_proof with two contexts (h2_init_devices, both on the one GPU here): every commit phase of keygen and
    create_proof is split by point range, the partial sums are added on the prover's device, and all five recorded proofs
    come out byte for byte (GWC and SHPLONK); reference surface: /root/reference/circuits/src/utils.rs:72-123, one
    create_proof call"""
    r = subprocess.run([sys.executable, "-c", _SHARDED_PROVER % ROOT], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sharded prover ok 5" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_streams_at_once_give_the_one_stream_results(h2):
    """MSMs on stream A and B interleaved with multi-pass NTTs on B and A: the arenas hand over with events"""
    import torch
    curve, n, m = "bn254", 1 << 12, 3
    cid = O.CURVE_IDS[curve]
    b = O.synth_bases(cid, SEED | 0xAB5, n).reshape(n, 8)
    bases = h2.Bases(curve, b)
    try:
        sets = [np.stack([O.synth_scalars(1, SEED | (0xA00 + 8 * s + j), n).reshape(n, 4) for j in range(m)]) for s in range(4)]
        devs = [torch.from_numpy(x.view(np.int64)).cuda() for x in sets]
        lg = 18                                                     # two passes: uses the NTT scratch arena
        big = [O.synth_scalars(1, SEED | (0xA80 + s), 1 << lg).reshape(1 << lg, 4) for s in range(2)]
        dbig = [torch.from_numpy(x.view(np.int64)).cuda() for x in big]
        w = _omega(curve, lg)
        outs = [torch.zeros((m, 12), dtype=torch.int64, device="cuda") for _ in range(4)]
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for rnd in range(2):
            bases.msm_device(devs[2 * rnd].data_ptr(), n, m, outs[2 * rnd].data_ptr(), sa.cuda_stream)
            h2.ntt_device(dbig[rnd].data_ptr(), 1, w, lg, curve, sb.cuda_stream)
            bases.msm_device(devs[2 * rnd + 1].data_ptr(), n, m, outs[2 * rnd + 1].data_ptr(), sb.cuda_stream)
        torch.cuda.synchronize()
        for s in range(4):
            res = outs[s].cpu().numpy().view(np.uint64)
            for j in range(m):
                assert np.array_equal(O.to_affine(cid, res[j]), O.to_affine(cid, O.best_multiexp(cid, sets[s][j], b))), (s, j)
        for s in range(2):
            assert np.array_equal(dbig[s].cpu().numpy().view(np.uint64), O.best_fft(1, big[s], w, lg, threads=8).reshape(-1, 4))
    finally:
        bases.release()


def test_six_streams_side_by_side_then_taking_over_each_others_scratch(h2):
    """every stream owns MSM / NTT scratch of its own for the first four streams (launch sequences on different streams
    run side by side); a fifth and sixth stream take over the slots idle longest, ordered behind their previous users"""
    import torch
    curve, n, m = "pallas", 1 << 13, 2
    cid = O.CURVE_IDS[curve]
    fid = O.CURVE_SCALAR_FIELD[cid]
    b = O.synth_bases(cid, SEED | 0xAB6, n).reshape(n, 8)
    bases = h2.Bases(curve, b)
    try:
        ns = 6
        streams = [torch.cuda.Stream() for _ in range(ns)]
        sets = [np.stack([O.synth_scalars(fid, SEED | (0xB00 + 8 * s + j), n).reshape(n, 4) for j in range(m)]) for s in range(ns)]
        devs = [torch.from_numpy(x.view(np.int64)).cuda() for x in sets]
        lg = 12                                                      # two passes: 6 + 6
        cols = [O.synth_scalars(fid, SEED | (0xB80 + s), 1 << lg).reshape(1 << lg, 4) for s in range(ns)]
        dcols = [torch.from_numpy(x.view(np.int64)).cuda() for x in cols]
        w = _omega(curve, lg)
        outs = [torch.zeros((3, m, 12), dtype=torch.int64, device="cuda") for _ in range(ns)]
        torch.cuda.synchronize()
        import ctypes
        lib = h2.load()
        before = (ctypes.c_uint64 * 4)()
        assert lib.h2_selftest_arena_stats(before) == 0
        for rnd in range(3):
            for s in range(ns):
                bases.msm_device(devs[s].data_ptr(), n, m, outs[s][rnd].data_ptr(), streams[s].cuda_stream)
                if rnd == 0:
                    h2.ntt_device(dcols[s].data_ptr(), 1, w, lg, curve, streams[s].cuda_stream)
        torch.cuda.synchronize()
        after = (ctypes.c_uint64 * 4)()
        assert lib.h2_selftest_arena_stats(after) == 0
        # six streams on four slots: slots change hands (with an event wait each time) for both kinds of scratch
        assert after[2] > before[2] and after[3] > before[3] and after[1] > before[1]
        for s in range(ns):
            want = [O.to_affine(cid, O.best_multiexp(cid, sets[s][j], b)) for j in range(m)]
            for rnd in range(3):
                res = outs[s][rnd].cpu().numpy().view(np.uint64)
                for j in range(m):
                    assert np.array_equal(O.to_affine(cid, res[j]), want[j]), (s, rnd, j)
            assert np.array_equal(dcols[s].cpu().numpy().view(np.uint64), O.best_fft(fid, cols[s], w, lg, threads=4).reshape(-1, 4)), s
    finally:
        bases.release()


def test_last_block_handoff_under_uneven_load(h2):
    """msm_final_kernel hands eight blocks' partials to the block that arrives last (counter + agent release / acquire).
    150 launches of the same MSM while another stream keeps every CU busy with multi-pass NTTs: every result must be the
    first one's point (a stale partial would show as a different point)."""
    import torch
    curve, n, m = "pallas", 1 << 14, 5
    cid = O.CURVE_IDS[curve]
    b = O.synth_bases(cid, SEED | 0xC15, n).reshape(n, 8)
    bases = h2.Bases(curve, b)
    try:
        cols = np.stack([O.synth_scalars(O.CURVE_SCALAR_FIELD[cid], SEED | (0xC40 + j), n).reshape(n, 4) for j in range(m)])
        dev = torch.from_numpy(cols.view(np.int64)).cuda()
        lg = 19
        dbig = torch.from_numpy(O.synth_scalars(O.CURVE_SCALAR_FIELD[cid], SEED | 0xC99, 3 << lg).reshape(3 << lg, 4).view(np.int64)).cuda()
        w = _omega(curve, lg)
        rounds = 150
        outs = torch.zeros((rounds, m, 12), dtype=torch.int64, device="cuda")
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for r in range(rounds):
            if r % 3 == 0:
                h2.ntt_device(dbig.data_ptr(), 3, w, lg, curve, sb.cuda_stream)
            bases.msm_device(dev.data_ptr(), n, m, outs[r].data_ptr(), sa.cuda_stream)
        torch.cuda.synchronize()
        res = outs.cpu().numpy().view(np.uint64)
        want = [O.to_affine(cid, res[0][j]) for j in range(m)]
        for j in range(m):
            assert np.array_equal(want[j], O.to_affine(cid, O.best_multiexp(cid, cols[j], b, threads=8))), j
        for r in range(1, rounds):
            for j in range(m):
                assert np.array_equal(O.to_affine(cid, res[r][j]), want[j]), (r, j)
    finally:
        bases.release()


def test_generate_params_on_a_side_stream_reproduces_the_pinned_file(h2):
    """the library calls of generate_params follow torch's CURRENT stream (their inputs are made there)"""
    import hashlib
    import torch
    from halo2_prover_amd import prover

    class Rng:
        def __init__(self):
            self.s = R.SurveyStream()

        def fill(self, n):
            return self.s.fill(n)

        def fr_random(self, _field=None):
            return self.s.fr_random(R.BN_FR)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        params = prover.generate_params(11, Rng())
    torch.cuda.synchronize()
    # sha256 of ParamsKZG::new(11).write() under the recorded RNG stream
    assert hashlib.sha256(params.write()).hexdigest() == "c071f033c580c8d827fb719c4d428a0d10673b4ab7da0c2ea62dec3ffc3fc6ca"


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no rank environment): the parent spawns two ranks before touching the GPU; on this
    one-GPU box they share cuda:0 and gather over gloo.  One JSON line, n_gpus = 2, sharded == unsharded."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--k", "12", "--no-proof", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["sharded_equals_unsharded"] is True and d["scaling"] == "weak" and d["one_proof_sharded"]["sharded_equals_unsharded"] is True


def test_transforms_queued_behind_an_msm_tail_give_the_same_results(h2):
    """h2_stream_wait_msm_tail: NTTs on a second stream start when the MSM's accumulate kernel has finished and run
    beside its tail kernels; bytes as on one stream"""
    import torch
    curve, n, m, lg = "bn254", 1 << 14, 3, 16
    cid = O.CURVE_IDS[curve]
    b = O.synth_bases(cid, SEED | 0xCB5, n).reshape(n, 8)
    bases = h2.Bases(curve, b)
    L = h2.load()
    try:
        cols = np.stack([O.synth_scalars(1, SEED | (0xC00 + j), n).reshape(n, 4) for j in range(m)])
        d = torch.from_numpy(cols.view(np.int64)).cuda()
        big = O.synth_scalars(1, SEED | 0xC80, 2 << lg).reshape(2, 1 << lg, 4)
        dbig = torch.from_numpy(big.view(np.int64)).cuda()
        w = _omega(curve, lg)
        out = torch.zeros((2, m, 12), dtype=torch.int64, device="cuda")
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for rnd in range(2):                       # the first wait only switches the marking on
            bases.msm_device(d.data_ptr(), n, m, out[rnd].data_ptr(), sa.cuda_stream)
            assert L.h2_stream_wait_msm_tail(sb.cuda_stream) == 0
            h2.ntt_device(dbig[rnd].data_ptr(), 1, w, lg, curve, sb.cuda_stream)
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint64)
        for rnd in range(2):
            for j in range(m):
                assert np.array_equal(O.to_affine(cid, res[rnd, j]), O.to_affine(cid, O.best_multiexp(cid, cols[j], b, threads=4)))
            assert np.array_equal(dbig[rnd].cpu().numpy().view(np.uint64), O.best_fft(1, big[rnd], w, lg, threads=8).reshape(-1, 4))
    finally:
        bases.release()


def test_one_launch_with_different_bases_per_column(h2):
    """h2_msm_device_multi: column j against its own registered bases (same length), incl. a sub-range"""
    import torch
    from halo2_prover_amd import api
    curve, n, m = "bn254", 3000, 4
    cid = O.CURVE_IDS[curve]
    ba = O.synth_bases(cid, SEED | 0xDB5, n).reshape(n, 8)
    bb = O.synth_bases(cid, SEED | 0xDB6, n).reshape(n, 8)
    A, B = h2.Bases(curve, ba), h2.Bases(curve, bb)
    try:
        cols = np.stack([O.synth_scalars(1, SEED | (0xD00 + j), n).reshape(n, 4) for j in range(m)])
        d = torch.from_numpy(cols.view(np.int64)).cuda()
        out = torch.zeros((m, 12), dtype=torch.int64, device="cuda")
        which = [A, B, B, A]
        api.msm_device_multi(which, d.data_ptr(), 0, n, n, out.data_ptr())
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint64)
        for j in range(m):
            want = O.best_multiexp(cid, cols[j], ba if which[j] is A else bb)
            assert np.array_equal(O.to_affine(cid, res[j]), O.to_affine(cid, want)), j
        lo, cnt = 1234, 1500
        api.msm_device_multi(which, d.data_ptr() + lo * 32, lo, cnt, n, out.data_ptr())
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint64)
        for j in range(m):
            bs = (ba if which[j] is A else bb)[lo:lo + cnt].copy()
            assert np.array_equal(O.to_affine(cid, res[j]), O.to_affine(cid, O.best_multiexp(cid, cols[j][lo:lo + cnt].copy(), bs))), j
        short = h2.Bases(curve, ba[:100])
        with pytest.raises(h2.H2Error):                    # different registered lengths cannot share a launch
            api.msm_device_multi([A, short], d.data_ptr(), 0, 100, n, out.data_ptr())
        short.release()
    finally:
        A.release()
        B.release()
