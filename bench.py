#!/usr/bin/env python3
"""bench.py -- MSM + NTT throughput of the halo2 hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--curve pallas|bn254|vesta] [--workload ...]

One "step" = one pass of the hot path over the MSM/NTT calls of ONE Poseidon k=16 proof
(BASELINE.json metric "MSM+NTT field-ops/s ... Poseidon k=16"; call counts from SURVEY.md
section 8(a)/App. A.4), with dense synthetic columns already resident in HBM:

    MSM  n = 2^16 : 4 advice + 2 permutation (over g_lagrange), 1 random-poly + 5 h pieces + 4 GWC
                    quotients (over g) = 16 MSMs, issued as Fiat-Shamir orders them: m = 4, 2 + 1 (the random
                    polynomial comes from the RNG, not the transcript: it shares the permutation products'
                    launch, with its own bases), 5, 4
    NTT           : 7 x iNTT(2^16), 7 x NTT(2^19), 1 x iNTT(2^19)   (extended domain, e = 3)

value = field-ops/s with the reference-parameter yardstick of SURVEY.md section 8(d):
ops_msm(n) = S*(11n + 32(2^c-1)) + 7*256 with c = ceil(ln n), S = 256/c+1;  ops_ntt(n) = 3*(n/2)*log2 n.
With N > 1 every rank (one per GPU) runs the whole step on columns of its own -- per-GPU work fixed, "scaling": "weak" --
and the commitment vector of every phase is all-gathered over RCCL (the path's one exchange step, m x 96 bytes per rank);
value = N steps' field-ops / step time.  ONE proof's job spread over the N GPUs (each MSM phase by whole columns when
m % N == 0, else a point range of every column per rank; NTT columns j -> rank j mod N) is the sub-record
"one_proof_sharded", checked against the unsharded commitments.

Printed keys beyond the driver contract: "roofline" (bucket-accumulate kernel, HIP-event timed
inside the library on the launch stream), "cpu_baseline" (the CPU oracle timed on this host,
rank 0, N = 1 only), "phases_ms", and the sub-records "bn254_step" (the same step on the curve whose parity is pinned
by the reference's recorded outputs), "drop_in_step" (the same step through the host-pointer entry points a patched
halo2_proofs would call, PCIe included), "headline_msm_2e20", "proof_gen".
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODULI = {
    "bn254_fr": 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    "pasta_fp": 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
    "pasta_fq": 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
}
SCALAR_FIELD = {"bn254": ("bn254_fr", 7, 28), "pallas": ("pasta_fq", 5, 32), "vesta": ("pasta_fp", 5, 32)}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
# multiply-adds of one mixed XYZZ addition in units of a full product's 117 (Pasta): 6 products + 2 squarings (81) + y3's
# two products with one reduction (198) = 1062 / 117
PRODUCT_EQUIVALENTS_PER_ADD = (6 * 117 + 2 * 81 + 198) / 117.0


def ops_msm(n):
    c = 1 if n < 4 else 3 if n < 32 else math.ceil(math.log(n))
    s = 256 // c + 1
    return s * (11 * n + 32 * ((1 << c) - 1)) + 7 * 256


def ops_ntt(n):
    return 3 * (n // 2) * int(math.log2(n))


def limbs(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def splitmix_columns(seed, n, p):
    """n field elements: SplitMix64 -> 4 limbs, top limb masked to 62 bits, one conditional subtract of
    the modulus.  The 256-bit patterns are used directly as the in-memory
    (Montgomery) representation: any value < p is a valid element."""
    idx = np.arange(1, 4 * n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    a = z.reshape(n, 4).copy()
    a[:, 3] &= np.uint64((1 << 62) - 1)
    pl = [(p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    # a >= p ?  (lexicographic from the top limb)
    ge = np.zeros(n, dtype=bool)
    eq = np.ones(n, dtype=bool)
    for i in (3, 2, 1, 0):
        ge |= eq & (a[:, i] > np.uint64(pl[i]))
        eq &= a[:, i] == np.uint64(pl[i])
    ge |= eq
    if ge.any():
        sub = a[ge]
        borrow = np.zeros(sub.shape[0], dtype=np.uint64)
        with np.errstate(over="ignore"):
            for i in range(4):
                pi = np.uint64(pl[i])
                d = sub[:, i] - pi - borrow
                borrow = ((sub[:, i] < pi) | ((sub[:, i] == pi) & (borrow == 1))).astype(np.uint64)
                sub[:, i] = d
        a[ge] = sub
    return a


def source_hash():
    """sha256 over the kernel sources: stamps profiles taken on THIS build (tools/pmc_traffic.py writes the same)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "halo2_prover_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no rank environment: start the N ranks as children, BEFORE this process has
    imported torch or touched a GPU (a process that has initialised the GPU is never re-exec'ed), relay rank 0's JSON
    line and exit with the children's status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        raise SystemExit("bench.py: the %d-rank run failed (exit %d)" % (args.gpus, proc.returncode))
    assert json.loads(line)["n_gpus"] == args.gpus
    print(line)


def jac_to_affine_ints(rows, q):
    """(m, 12) uint64 Jacobian points in Montgomery form -> list of affine (x, y) ints (None = identity), for the
    sharded-vs-whole comparison (group elements, not representatives)"""
    rinv = pow(1 << 256, -1, q)
    out = []
    for r in rows:
        X, Y, Z = (sum(int(r[4 * i + j]) << (64 * j) for j in range(4)) * rinv % q for i in range(3))
        if Z == 0:
            out.append(None)
            continue
        zi = pow(Z, -1, q)
        out.append((X * zi * zi % q, Y * zi * zi * zi % q))
    return out


BASE_FIELD = {"bn254": 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
              "pallas": MODULI["pasta_fp"], "vesta": MODULI["pasta_fq"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--curve", default="pallas", choices=["pallas", "bn254", "vesta"])
    ap.add_argument("--k", type=int, default=16, help="rows = 2^k (Poseidon k=16 is the metric's config)")
    ap.add_argument("--workload", default="poseidon", choices=["poseidon", "msm", "ntt"],
                    help="poseidon = the proof-shaped MSM+NTT mix (default); msm / ntt = one kernel family only")
    ap.add_argument("--msm-cols", type=int, default=1, help="columns per launch for --workload msm")
    ap.add_argument("--ntt-cols", type=int, default=1, help="columns per launch for --workload ntt")
    ap.add_argument("--schedule", default="early_tail", choices=["tail", "free", "early", "early_tail", "early_split", "serial"],
                    help="where the step's challenge-free transforms run: beside the MSM tails (default), unordered on a "
                         "second stream, advice transforms from the step's start, or all on the main stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-proof", action="store_true", help="skip the end-to-end Poseidon proof (proof-gen ms)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sub-records: modmul ceiling, 2^20 headline MSM (config 4), config 5")
    ap.add_argument("--config5-k", type=int, default=24, help="rows of the config-5 sub-record (N > 1 only)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="rehearsal: take the N > 1 code paths (range split, all-gather, partial-sum addition, config 5) even "
                         "with one rank -- run under torch.distributed.run with one rank to exercise RCCL on a one-GPU box")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, sys.argv[1:])

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    device_index = local_rank % ndev
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    use_dist = "RANK" in os.environ                    # under torch.distributed.run the collective path is exercised
    backend = None                                     # even with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL needs one GPU per rank; on a box with fewer (rehearsals on the one-GPU box) the ranks share a GPU
        # and the 96-byte results are gathered through host memory with gloo
        backend = "nccl" if ndev >= world else "gloo"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus

    import halo2_prover_amd as h2
    from halo2_prover_amd import lib as h2lib
    from halo2_prover_amd import sharded
    h2.init(device_index)
    L = h2.load()
    cid = h2.CURVES[args.curve]
    fname, gen, two_adicity = SCALAR_FIELD[args.curve]
    p = MODULI[fname]
    R = (1 << 256) % p
    k = args.k
    n = 1 << k
    ext = 3                                   # Poseidon: degree 6 -> extended_k = k + 3
    # an explicit (non-NULL) stream: the library maps a NULL stream argument to its own stream, and the
    # all-gather below must be ordered after the MSM launches, so everything runs on this one
    work_stream = torch.cuda.Stream(device=dev, priority=-1)   # the commit phases: ahead of the side stream's transforms
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream
    assert stream != 0

    def to_dev(a):
        return torch.from_numpy(a.view(np.int64)).to(dev)

    def omega(log_n, inverse=False):
        root = pow(gen, (p - 1) >> two_adicity, p)
        w = pow(root, 1 << (two_adicity - log_n), p)
        if inverse:
            w = pow(w, -1, p)
        return limbs(w * R % p)

    def make_srs(count, sval):
        """[s^i]G for i < count, made on the device (valid curve points without the CPU oracle; the survey's
        try-and-increment points cost the same to add), registered as resident bases"""
        buf = torch.empty((count, 8), dtype=torch.int64, device=dev)
        s_m = limbs(sval * R % p)
        h2lib.check(L.h2_srs_generate(cid, s_m.ctypes.data, count, buf.data_ptr(), stream), "h2_srs_generate")
        torch.cuda.synchronize()
        b = h2.Bases.from_device(args.curve, buf.data_ptr(), count)
        del buf
        return b

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- resident inputs ---------------------------------------------------------------------------------------------
    seed = 0x48324D5300000000
    g_lagrange, g = make_srs(n, 0x1234567), make_srs(n, 0x7654321)
    plan = g.plan()

    # the commit phases of one proof as the transcript orders them.  The vanishing argument's random polynomial is
    # drawn from the RNG, not from the transcript, so its commitment (over g) shares the launch of the permutation
    # products (over g_lagrange): 16 MSMs in 4 launch sequences, m = 4, 2 + 1, 5, 4
    phases = [(g_lagrange, 4), ([g_lagrange, g_lagrange, g], 3), (g, 5), (g, 4)] if args.workload == "poseidon" else \
        [(g, args.msm_cols)] if args.workload == "msm" else []
    n_msm = sum(m for _, m in phases)
    cols_np = splitmix_columns(seed | 1, max(n_msm, 1) * n, p)
    msm_cols = to_dev(cols_np)                # (n_msm*n, 4); the same columns on every rank (32 MiB at k = 16): the
    # one-proof sub-record needs them identical, and the values do not change what a column costs
    # NTT groups of the step, with what they depend on in a real proof (prover.py / SURVEY.md App. A.4): the
    # Lagrange -> coefficient -> extended-coset transforms of the advice + instance columns need only the witness, so
    # they run on a second stream while the commit phases' MSMs run on the first; the two permutation products' start
    # when phase 2's inputs exist; the quotient's inverse transform needs challenge y, i.e. everything before phase 4
    if args.workload == "poseidon":
        ntts = [("advice_i", k, 5, True, "beside tail 0"), ("advice_e_a", k + ext, 2, False, "beside tail 0"),
                ("advice_e_b", k + ext, 3, False, "beside tail 1"),
                ("z_i", k, 2, True, "beside tail 1"), ("z_e", k + ext, 2, False, "beside tail 1"),
                ("h_i", k + ext, 1, True, "main stream, before the quotient's commitment")]
    elif args.workload == "ntt":
        ntts = [("cols", k, args.ntt_cols, False, "main")]
    else:
        ntts = []
    multi = world > 1 or (args.force_collectives and use_dist)

    def build_ntt_bufs(shard):
        bufs = {}
        for j, (name, lg, m, inv, _) in enumerate(ntts):
            # the one-proof job (sub-record, below) shards NTT columns whole: column j -> rank j mod N, no collective
            mine = list(range(rank, m, world)) if shard else list(range(m))
            if mine:
                allc = splitmix_columns(seed | (2 + j) | (0 if shard else rank << 16), m << lg, p).reshape(m, 1 << lg, 4)
                bufs[name] = (to_dev(np.ascontiguousarray(allc[mine]).reshape(-1, 4)), lg, len(mine), omega(lg, inv))
        return bufs

    # N > 1, the line's `value` (weak scaling: per-GPU work fixed): every rank runs the WHOLE proof-shaped step on columns
    # of its own -- N proofs' worth of commitments and transforms on N GPUs -- and the commitment vector of every phase is
    # all-gathered (the path's one exchange step: m x 96 bytes per rank and phase).  The north star's other reading, ONE
    # proof's columns spread over the N GPUs (total work fixed), is the sub-record `one_proof_sharded`.
    ntt_bufs = build_ntt_bufs(False)
    ntt_bufs_shared = build_ntt_bufs(True) if world > 1 else ntt_bufs
    results = [None] * len(phases)

    class Lane:
        """the streams and events of one step in flight: the commit phases' stream, the transforms' second stream"""
        def __init__(self, work, bufs):
            self.work_stream, self.stream = work, work.cuda_stream
            self.side_stream = torch.cuda.Stream(device=dev, priority=0)
            self.side = self.side_stream.cuda_stream
            self.ev_start, self.ev_side_done, self.ev_phase0 = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            self.results = [None] * len(phases)
            self.bufs = bufs

    lane0 = Lane(work_stream, None)
    lane0.results = results
    side_stream, side = lane0.side_stream, lane0.side

    def run_phase(i, off, mode="single", gather=False, lane=lane0):
        bases, m = phases[i]
        lane.results[i] = sharded.msm_phase_device(bases, msm_cols.data_ptr() + off * n * 32, n, m, lane.stream,
                                                   mode=mode if multi else "single")
        if gather and multi:
            sharded.all_gather_rows(lane.results[i])        # every rank holds every rank's commitments of the phase
        return off + m

    def msm_phase(mode="single"):
        off = 0
        for i in range(len(phases)):
            off = run_phase(i, off, mode)

    def run_ntt(name, st, bufs=None):
        bufs = ntt_bufs if bufs is None else bufs
        if name in bufs:
            buf, lg, m, w = bufs[name]
            h2.ntt_device(buf.data_ptr(), m, w, lg, args.curve, st)

    def ntt_phase():
        for name, *_ in ntts:
            run_ntt(name, stream)

    def step(shared=False, lane=lane0, sched=None):
        """shared = False: this rank's own step (mode "single", commitments gathered).  shared = True: one proof's job
        spread over the ranks (columns when m % N == 0, else point ranges; NTT columns j -> rank j mod N).
        `lane`: the streams the step is enqueued on (torch's current stream must be lane.work_stream)."""
        mode = (None if world > 1 else "range") if shared else "single"
        bufs = ntt_bufs_shared if shared else (lane.bufs or ntt_bufs)
        stream, side, side_stream, work_stream = lane.stream, lane.side, lane.side_stream, lane.work_stream
        if args.workload != "poseidon":
            off = 0
            for i in range(len(phases)):
                off = run_phase(i, off, mode, not shared, lane)
            for name, *_ in ntts:
                run_ntt(name, stream, bufs)
            return
        # first stream: the commit phases in Fiat-Shamir order.  Second stream: the transforms that wait for no
        # challenge.  Default placement ("early_tail"): the advice columns exist before their commitment does, so their
        # transforms are queued from the step's start; those of the permutation products wait for the accumulate kernel
        # of the products' own commit phase (h2_stream_wait_msm_tail) and run beside its small-grid tail
        sched = sched or args.schedule
        if sched == "serial":
            off = run_phase(0, 0, mode, not shared, lane)
            off = run_phase(1, off, mode, not shared, lane)
            for name, *_ in ntts:
                run_ntt(name, stream, bufs)
            off = run_phase(2, off, mode, not shared, lane)
            run_phase(3, off, mode, not shared, lane)
            return
        lane.ev_start.record(work_stream)
        side_stream.wait_event(lane.ev_start)
        early = sched.startswith("early")
        if early:
            run_ntt("advice_i", side, bufs)
            run_ntt("advice_e_a", side, bufs)
            if sched != "early_split":
                run_ntt("advice_e_b", side, bufs)
        off = run_phase(0, 0, mode, not shared, lane)  # advice
        if sched in ("tail", "early_split"):
            L.h2_stream_wait_msm_tail(side)
        if sched in ("tail", "free"):
            run_ntt("advice_i", side, bufs)
            run_ntt("advice_e_a", side, bufs)
        if sched == "early_split":
            run_ntt("advice_e_b", side, bufs)
        if sched in ("free", "early", "early_split"):   # z exists once beta, gamma (phase 0's commitments) do
            lane.ev_phase0.record(work_stream)
            side_stream.wait_event(lane.ev_phase0)
        off = run_phase(1, off, mode, not shared, lane)  # permutation products (exist once beta, gamma do) + random polynomial
        if sched in ("tail", "early_tail"):
            L.h2_stream_wait_msm_tail(side)
        if not early:
            run_ntt("advice_e_b", side, bufs)
        run_ntt("z_i", side, bufs)
        run_ntt("z_e", side, bufs)
        lane.ev_side_done.record(side_stream)
        work_stream.wait_event(lane.ev_side_done)    # y is squeezed next; the quotient needs every extended column
        run_ntt("h_i", stream, bufs)
        off = run_phase(2, off, mode, not shared, lane)  # quotient pieces
        run_phase(3, off, mode, not shared, lane)        # opening witnesses

    for _ in range(args.warmup):
        step()
    barrier()
    L.h2_profile_enable(1)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    prof = h2lib.Profile()
    h2lib.check(L.h2_profile_read(ctypes.byref(prof)), "h2_profile_read")
    L.h2_profile_enable(0)
    if use_dist and multi:
        t = torch.tensor([dt], dtype=torch.float64)
        if backend == "nccl":
            t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # one proof's job spread over the ranks (total work fixed): sharded == whole as group elements, then its time
    sharded_ok = None
    one_proof = None
    if multi and phases:
        q = BASE_FIELD[args.curve]
        step(shared=True)
        torch.cuda.synchronize()
        got = [jac_to_affine_ints(r.cpu().numpy().view(np.uint64), q) for r in results]
        msm_phase(mode="single")
        torch.cuda.synchronize()
        want = [jac_to_affine_ints(r.cpu().numpy().view(np.uint64), q) for r in results]
        sharded_ok = got == want
        if not sharded_ok:       # recorded in the line (sharded_equals_unsharded: false), not raised: a sub-record's
            # failure must not take the measured line with it -- or leave the other ranks waiting in a collective
            print("rank %d: SHARDED COMMITMENTS DIFFER from the unsharded ones" % rank, file=sys.stderr)
        barrier()
        ts = time.perf_counter()
        for _ in range(args.steps):
            step(shared=True)
        barrier()
        dts = time.perf_counter() - ts
        if use_dist:
            t = torch.tensor([dts], dtype=torch.float64)
            if backend == "nccl":
                t = t.to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        one_proof = {"what": "ONE proof-shaped step spread over the %d ranks (total work fixed): each MSM phase by whole "
                             "columns when m %% N == 0, else a point range of every column per rank; one all-gather of "
                             "96-byte points per phase; NTT columns j -> rank j mod N" % world,
                     "ms_per_step": round(dts / args.steps * 1e3, 4), "sharded_equals_unsharded": sharded_ok}

    # per-phase timing (outside the timed region; torch events see this stream because the library was
    # handed torch's current stream)
    phases_ms = {}
    prof_alone = None
    for name, fn in (("msm", msm_phase), ("ntt", ntt_phase)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()                                    # (scratch sized for this stream's calls before the clock starts)
        barrier()
        if name == "msm" and phases:
            L.h2_profile_enable(1)              # the accumulate kernel with nothing beside it (roofline: *_alone)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if name == "msm" and phases:
            prof_alone = h2lib.Profile()
            h2lib.check(L.h2_profile_read(ctypes.byref(prof_alone)), "h2_profile_read")
            L.h2_profile_enable(0)
        phases_ms[name] = round(e0.elapsed_time(e1) / 10, 4)

    # two steps in flight on the one GPU (sub-record; the line's value keeps one step at a time): even steps on the
    # first pair of streams, odd steps on a second pair with transform buffers of its own -- the library gives every
    # stream its own MSM / NTT scratch -- so that one step's sorts and small-grid tails run beside the other's
    # chip-filling kernels.  Each step keeps its own Fiat-Shamir order.
    two_in_flight = None
    if args.workload == "poseidon" and world == 1 and not args.no_extras:
        # (normal priority: a SECOND high-priority stream in the process left every later small kernel of the run 3-5 x
        # slower -- 50 us for a 13 us kernel; profiles/r03_second_high_priority_stream.txt)
        lane1 = Lane(torch.cuda.Stream(device=dev, priority=0),
                     {k_: (b.clone(), lg, m, w) for k_, (b, lg, m, w) in ntt_bufs.items()})
        lanes = [lane0, lane1]

        def flight(count):
            for i in range(count):
                ln = lanes[i & 1]
                with torch.cuda.stream(ln.work_stream):
                    step(lane=ln, sched="early")
        flight(4)
        torch.cuda.synchronize()
        # the second lane computes what the first does (same inputs): its commitments must be the first lane's
        qf = BASE_FIELD[args.curve]
        same = all(jac_to_affine_ints(a.cpu().numpy().view(np.uint64), qf) == jac_to_affine_ints(b.cpu().numpy().view(np.uint64), qf)
                   for a, b in zip(lane0.results, lane1.results))
        if not same:
            print("two steps in flight: THE LANES' COMMITMENTS DIFFER (lanes_agree: false in the line)", file=sys.stderr)
        cnt = 2 * max(args.steps, 10)
        tf = time.perf_counter()
        flight(cnt)
        torch.cuda.synchronize()
        dtf = time.perf_counter() - tf
        two_in_flight = {"what": "two proof-shaped steps in flight on one GPU (alternating stream pairs, each with its own "
                                 "scratch); throughput over %d steps" % cnt,
                         "ms_per_step": round(dtf / cnt * 1e3, 4), "steps_per_s": cnt / dtf,
                         "lanes_agree": same}

    overlap = None
    if args.workload == "poseidon":
        overlap = {"ms_per_step": round(dt / args.steps * 1e3, 4), "msm_plus_ntt_serial_ms": round(phases_ms["msm"] + phases_ms["ntt"], 4),
                   "what": "phases_ms times the MSM phases and the NTTs alone, one after the other on one stream; in the "
                           "step the transforms that wait for no challenge run on a second stream beside the MSMs' "
                           "small-grid tails (the Fiat-Shamir order of the commit phases is kept)"}

    # NTT roofline: algorithmic bytes = m * n * 64 per transform (SURVEY.md 8(d) bytes_ntt: each element read and
    # written once, whatever the number of passes); the per-pass figure the kernels actually move is reported beside it
    roofline_ntt = None
    if ntt_bufs:
        ntt_bytes = sum(m * (1 << lg) * 64 for _, lg, m, _ in ntt_bufs.values())
        pass_bytes = sum(m * (1 << lg) * 64 * ((lg + 9) // 10) for _, lg, m, _ in ntt_bufs.values())
        ach = ntt_bytes / (phases_ms["ntt"] * 1e-3) / 1e9
        roofline_ntt = {"bound": "hbm", "kernel": "ntt29_pass_kernel (all launches of this rank's NTTs of the step)",
                        "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                        "algorithmic_bytes_per_step": ntt_bytes, "per_pass_bytes_per_step": pass_bytes,
                        "ms_per_step": phases_ms["ntt"]}

    ops_step = n_msm * ops_msm(n) + sum(m * ops_ntt(1 << lg) for _, lg, m, _, _ in ntts)
    if two_in_flight:
        two_in_flight["field_ops_per_s"] = ops_step * two_in_flight.pop("steps_per_s")
    value = world * ops_step * args.steps / dt     # every rank ran the step on its own columns: N steps' field-ops per step time

    # HBM-side traffic of the dominant kernel comes from a separate rocprofv3 --pmc run (counters cannot be read
    # in-process); the committed summary is used only when it was taken on THIS build of the kernels
    traffic = ntt_traffic = traffic_source = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if (args.workload == "poseidon" and args.curve == "pallas" and k == 16 and world == 1
                and pmc.get("source_hash") == source_hash()):
            traffic = pmc["dominant_kernel"]["traffic_bytes_per_launch"]
            per = next(v for name, v in pmc["kernels"].items() if "pass_kernel" in name and "ntt" in name)
            ntt_traffic = per["traffic_bytes_per_launch"] * per["launches"] // pmc["steps_profiled"]
            traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc, kernel sources %s, head %s)" % (
                pmc["source_hash"], pmc.get("git_head", "?"))
    except Exception:
        traffic = ntt_traffic = traffic_source = None
    if roofline_ntt:
        roofline_ntt["traffic"] = ntt_traffic          # bytes per step, like algorithmic_bytes_per_step

    # the integer ceiling: dependent working-form products on every CU (a few ms), at the chunk kernel's occupancy
    # and at the best occupancy
    modmul = None
    if not args.no_extras:
        r3, r8 = ctypes.c_double(0), ctypes.c_double(0)
        h2lib.check(L.h2_selftest_modmul_rate(cid, 3, 512, ctypes.byref(r3)), "h2_selftest_modmul_rate")
        h2lib.check(L.h2_selftest_modmul_rate(cid, 8, 512, ctypes.byref(r8)), "h2_selftest_modmul_rate")
        modmul = {"unit": "modmul/s", "at_3_waves_per_simd": r3.value, "at_8_waves_per_simd": r8.value,
                  "what": "dependent 9 x 29-bit Montgomery products of the base field on every CU, measured in this run"}

    roofline = None
    if prof.launches:
        achieved = prof.algorithmic_bytes / (prof.kernel_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "msm_chunk_kernel", "achieved": round(achieved, 3),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "avg_kernel_ms": round(prof.kernel_ms / prof.launches, 5), "launches": int(prof.launches),
                    "algorithmic_bytes_per_launch": round(prof.algorithmic_bytes / prof.launches, 1),
                    "note": "the contract's HBM fraction; the kernel is bound by the integer VALU, see modmul_frac. "
                            "Events over the timed region: in the step the first commit phase's accumulate kernel shares "
                            "the chip with the advice columns' transforms (--schedule early_tail); *_alone = the same "
                            "events over the MSM launch sequences run by themselves (phases_ms.msm)"}
        if prof_alone is not None and prof_alone.launches:
            roofline["avg_kernel_ms_alone"] = round(prof_alone.kernel_ms / prof_alone.launches, 5)
            roofline["frac_alone"] = round(prof_alone.algorithmic_bytes / (prof_alone.kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
        if modmul:
            # one mixed addition (8M + 2S = 10 products, the formula's nominal count) per sorted entry; entries = terms
            # x windows.  Since round 3 the kernel executes fewer multiply-adds than ten full products: the two squarings
            # take 81 instead of 117 of them and y3's two products share one reduction (198 instead of 234): 9.07
            # product-equivalents per addition -- modmul_frac keeps the nominal 10 (comparable across rounds),
            # modmul_frac_executed prices what is executed
            terms = prof.algorithmic_bytes / 96.0
            mm = terms * plan["windows"] * 10
            roofline["modmul_per_s"] = mm / (prof.kernel_ms * 1e-3)
            roofline["modmul_frac"] = round(roofline["modmul_per_s"] / modmul["at_3_waves_per_simd"], 4)
            roofline["modmul_frac_executed"] = round(roofline["modmul_frac"] * PRODUCT_EQUIVALENTS_PER_ADD / 10.0, 4)
            if prof_alone is not None and prof_alone.launches:
                mm_alone = prof_alone.algorithmic_bytes / 96.0 * plan["windows"] * 10 / (prof_alone.kernel_ms * 1e-3)
                roofline["modmul_frac_alone"] = round(mm_alone / modmul["at_3_waves_per_simd"], 4)
                roofline["modmul_frac_executed_alone"] = round(roofline["modmul_frac_alone"] * PRODUCT_EQUIVALENTS_PER_ADD / 10.0, 4)
            if "msm" in phases_ms and n_msm:
                whole = (n_msm * n * plan["windows"] * 10) / (phases_ms["msm"] * 1e-3)     # this rank's own 16 columns
                roofline["msm_phase_modmul_per_s"] = whole
                roofline["msm_phase_modmul_frac"] = round(whole / modmul["at_8_waves_per_simd"], 4)

    def sub(fn, *a):
        """a sub-record's failure is recorded in the line, it does not take the measured values with it.  (With several
        ranks this holds for failures every rank meets at the same place; a rank that fails alone still leaves the others
        in the sub-record's next collective.)"""
        try:
            return fn(*a)
        except Exception as exc:  # noqa: BLE001
            print("sub-record %s failed: %r" % (fn.__name__, exc), file=sys.stderr)
            return {"error": repr(exc)[:400]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = sub(cpu_baseline, args, cols_np, n, k, ext, p, gen, two_adicity, R)

    extras = {}
    if not args.no_extras and args.workload == "poseidon":
        for b in (g, g_lagrange):
            b.release()
        del msm_cols
        ntt_bufs.clear()
        torch.cuda.empty_cache()
        extras["headline_msm_2e20"] = sub(headline_msm, args, 20, make_srs, world, rank, dev, stream, barrier, modmul, p, multi)
        if multi:
            extras["config5"] = sub(config5, args, make_srs, world, rank, dev, stream, barrier, p, gen, two_adicity, R)
        if world == 1 and not multi:
            # sub-records on one GPU: the pinned curve, and the host-pointer route
            if args.curve != "bn254":
                extras["bn254_step"] = sub(step_on_curve, args, "bn254", dev)
            extras["drop_in_step"] = sub(drop_in_step, args, args.curve)

    proof_gen = None
    # h2_generate_proof is one process on this rank's GPU(s): rank 0 runs and reports it; n_gpus in the record says what
    # ran (the C++ prover shards its commit phases over the contexts of h2_init_devices, not over ranks)
    proof_gen_multi = None
    if not args.no_proof and args.workload == "poseidon" and rank == 0:
        proof_gen = sub(proof_generation, k)
    if not args.no_proof and args.workload == "poseidon" and world > 1 and ndev >= world:
        # the N-GPU proof: ONE process with N contexts (h2_init_devices), started by rank 0 while the other ranks wait at
        # the barrier below with their GPUs idle; its commit phases are split by point range over the N GPUs
        if rank == 0:
            import subprocess
            # a sub-record must never cost the line: bounded in time, every failure recorded instead of raised
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "proof_bench.py"), str(k), str(world)],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                proof_gen_multi = json.loads(line[-1]) if r.returncode == 0 and line else {"error": r.stderr[-400:]}
            except subprocess.TimeoutExpired:
                proof_gen_multi = {"error": "tools/proof_bench.py %d %d did not finish within 240 s" % (k, world)}
            except Exception as exc:  # noqa: BLE001
                proof_gen_multi = {"error": repr(exc)[:400]}
        barrier()

    if rank == 0:
        workload = {"poseidon": "poseidon_k%d_proof_shape: 16 MSM(2^%d) in launches m=4,2+1,5,4 + 7 iNTT(2^%d) + "
                                "7 NTT(2^%d) + 1 iNTT(2^%d)" % (k, k, k, k + ext, k + ext),
                    "msm": "msm(2^%d) x %d columns" % (k, args.msm_cols), "ntt": "ntt(2^%d) x %d columns" % (k, args.ntt_cols)}[args.workload]
        if world == 1:
            par = "1 GPU"
        else:
            par = ("%d ranks (%s), one per GPU: every rank runs the whole proof-shaped step on columns of its own (per-GPU "
                   "work fixed), the commitment vector of every phase is all-gathered (m x 96 bytes per rank); the "
                   "one-proof split is the sub-record one_proof_sharded" % (world, "RCCL" if backend == "nccl" else
                                                                            "gloo: %d ranks share %d GPU(s)" % (world, ndev)))
        out = {
            "metric": "MSM+NTT field-ops/s", "value": value, "unit": "field-ops/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_median": per_step[len(per_step) // 2], "ms_per_step_min": per_step[0],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (256-bit modular)",
            "data": "synthetic (dense SplitMix64 scalars as SURVEY 8(d) defines them; bases [s^i]G made on the device instead of "
                    "8(d)'s try-and-increment points: the same cost per addition, a different distribution)",
            "parity": "bn254 pinned by the reference's recorded params / proofs; pallas and vesta self-consistent "
                      "(no reference vector exists)",
            "config": {"workload": workload, "curve": args.curve, "k": k, "columns": n_msm,
                       "parallelism": par, "backend": backend,
                       "msm_window_bits": plan["window_bits"], "msm_windows": plan["windows"],
                       "msm_table_bytes": plan["table_bytes"]},
            "sharded_equals_unsharded": sharded_ok, "one_proof_sharded": one_proof,
            "roofline": roofline, "roofline_ntt": roofline_ntt, "modmul_ceiling": modmul,
            "cpu_baseline": cpu, "proof_gen": proof_gen, "proof_gen_n_gpus": proof_gen_multi,
            "phases_ms": phases_ms, "overlap": overlap, "two_steps_in_flight": two_in_flight, "field_ops_per_step": ops_step,
        }
        out.update(extras)
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _device_scalars(L, cid, count, seed_byte, dev, stream):
    """`count` uniformly random scalars made on the device (ChaCha20 draws reduced mod the field)"""
    import torch
    from halo2_prover_amd import lib as h2lib
    t = torch.empty((count, 4), dtype=torch.int64, device=dev)
    seed = bytes([seed_byte]) * 32
    h2lib.check(L.h2_chacha20_scalars_device(cid, seed, 0, count, t.data_ptr(), stream), "h2_chacha20_scalars_device")
    return t


def step_on_curve(args, curve, dev, steps=10, warmup=2):
    """The Poseidon k = 16 proof-shaped step on another curve, one GPU, as a sub-record: same launches (m = 4, 2 + 1, 5,
    4), same transforms, same two streams as the headline step; timed with the contract's bracket."""
    import torch
    import halo2_prover_amd as h2
    from halo2_prover_amd import lib as h2lib
    from halo2_prover_amd import sharded
    L = h2.load()
    cid = h2.CURVES[curve]
    fname, gen, two_adicity = SCALAR_FIELD[curve]
    p = MODULI[fname]
    R = (1 << 256) % p
    k, ext = args.k, 3
    n = 1 << k
    work_stream = torch.cuda.current_stream()
    stream = work_stream.cuda_stream
    side_stream = torch.cuda.Stream(device=dev, priority=0)
    side = side_stream.cuda_stream

    def srs(sval):
        buf = torch.empty((n, 8), dtype=torch.int64, device=dev)
        h2lib.check(L.h2_srs_generate(cid, limbs(sval * R % p).ctypes.data, n, buf.data_ptr(), stream), "h2_srs_generate")
        torch.cuda.synchronize()
        b = h2.Bases.from_device(curve, buf.data_ptr(), n)
        del buf
        return b

    def omega(log_n, inverse=False):
        root = pow(gen, (p - 1) >> two_adicity, p)
        w = pow(root, 1 << (two_adicity - log_n), p)
        return limbs((pow(w, -1, p) if inverse else w) * R % p)

    g_lagrange, g = srs(0x1234567), srs(0x7654321)
    phases = [(g_lagrange, 4), ([g_lagrange, g_lagrange, g], 3), (g, 5), (g, 4)]
    cols = torch.from_numpy(splitmix_columns(0x48324D5300000101, 16 * n, p).view(np.int64)).to(dev)
    ntts = [("advice_i", k, 5, True), ("advice_e_a", k + ext, 2, False), ("advice_e_b", k + ext, 3, False),
            ("z_i", k, 2, True), ("z_e", k + ext, 2, False), ("h_i", k + ext, 1, True)]
    bufs = {name: (torch.from_numpy(splitmix_columns(0x48324D5300000110 + j, m << lg, p).view(np.int64)).to(dev), lg, m,
                   omega(lg, inv)) for j, (name, lg, m, inv) in enumerate(ntts)}
    ev_start, ev_side_done, ev_phase0 = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()

    def phase(i, off):
        bases, m = phases[i]
        sharded.msm_phase_device(bases, cols.data_ptr() + off * n * 32, n, m, stream, mode="single")
        return off + m

    def ntt(name, st):
        buf, lg, m, w = bufs[name]
        h2.ntt_device(buf.data_ptr(), m, w, lg, curve, st)

    def step():                                  # the headline step's default placement ("early_tail")
        ev_start.record(work_stream)
        side_stream.wait_event(ev_start)
        ntt("advice_i", side)
        ntt("advice_e_a", side)
        ntt("advice_e_b", side)
        off = phase(0, 0)
        off = phase(1, off)
        L.h2_stream_wait_msm_tail(side)
        ntt("z_i", side)
        ntt("z_e", side)
        ev_side_done.record(side_stream)
        work_stream.wait_event(ev_side_done)
        ntt("h_i", stream)
        off = phase(2, off)
        phase(3, off)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    L.h2_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = h2lib.Profile()
    h2lib.check(L.h2_profile_read(ctypes.byref(prof)), "h2_profile_read")
    L.h2_profile_enable(0)
    r3 = ctypes.c_double(0)
    h2lib.check(L.h2_selftest_modmul_rate(cid, 3, 512, ctypes.byref(r3)), "h2_selftest_modmul_rate")
    plan = g.plan()
    ops_step = 16 * ops_msm(n) + sum(m * ops_ntt(1 << lg) for _, lg, m, _ in ntts)
    mm = prof.algorithmic_bytes / 96.0 * plan["windows"] * 10 / (prof.kernel_ms * 1e-3)
    rec = {"curve": curve, "workload": "the same step (16 MSM(2^%d) + 15 NTT) on %s" % (k, curve), "steps": steps,
           "ms_per_step": round(dt / steps * 1e3, 4), "field_ops_per_s": ops_step * steps / dt,
           "accumulate_kernel_ms": round(prof.kernel_ms / prof.launches, 5),
           "accumulate_hbm_frac": round(prof.algorithmic_bytes / (prof.kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
           "modmul_ceiling_at_3_waves": r3.value, "modmul_frac": round(mm / r3.value, 4),
           "modmul_frac_executed": round(mm / r3.value * PRODUCT_EQUIVALENTS_PER_ADD / 10.0, 4),
           "parity": "pinned: params files, proofs and vk digests recorded from the reference's own build (tests/golden)"}
    for b in (g, g_lagrange):
        b.release()
    del cols, bufs
    torch.cuda.empty_cache()
    return rec


def drop_in_step(args, curve, steps=5):
    """What a maintainer gets from INTEGRATION.md section 3 alone: the same 16 MSMs and 15 NTTs, one call each, through
    the HOST-pointer entry points best_multiexp / best_fft would be patched to call (h2_msm, h2_ntt) on pageable host
    columns -- H2D of every column, the launch sequence, D2H of the result, per call.  Wall clock."""
    import halo2_prover_amd as h2
    L = h2.load()
    cid = h2.CURVES[curve]
    fname, gen, two_adicity = SCALAR_FIELD[curve]
    p = MODULI[fname]
    R = (1 << 256) % p
    k, ext = args.k, 3
    n = 1 << k
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    buf = torch.empty((n, 8), dtype=torch.int64, device=dev)
    L.h2_srs_generate(cid, limbs(0x1234567 * R % p).ctypes.data, n, buf.data_ptr(), None)
    torch.cuda.synchronize()
    host_bases = buf.cpu().numpy().view(np.uint64)
    del buf
    bases = h2.Bases(curve, host_bases)               # h2_bases_register: once per SRS, as the shim does
    root = pow(gen, (p - 1) >> two_adicity, p)

    def omega(lg):
        return limbs(pow(root, 1 << (two_adicity - lg), p) * R % p)

    cols = [splitmix_columns(0x48324D5300000201 + j, n, p) for j in range(16)]
    small = [splitmix_columns(0x48324D5300000221 + j, n, p) for j in range(7)]
    big = [splitmix_columns(0x48324D5300000231 + j, n << ext, p) for j in range(8)]
    out = np.zeros(12, dtype=np.uint64)

    def step():
        for c in cols:
            L.h2_msm(cid, bases.handle, c.ctypes.data, n, out.ctypes.data)
        for a in small:
            L.h2_ntt(cid, a.ctypes.data, omega(k).ctypes.data, k)
        for a in big:
            L.h2_ntt(cid, a.ctypes.data, omega(k + ext).ctypes.data, k + ext)

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    t1 = time.perf_counter()
    for c in cols:
        L.h2_msm(cid, bases.handle, c.ctypes.data, n, out.ctypes.data)
    t_msm = (time.perf_counter() - t1) / 16
    t2 = time.perf_counter()
    for a in big:
        L.h2_ntt(cid, a.ctypes.data, omega(k + ext).ctypes.data, k + ext)
    t_big = (time.perf_counter() - t2) / 8
    ops_step = 16 * ops_msm(n) + 7 * ops_ntt(n) + 8 * ops_ntt(n << ext)
    bases.release()
    return {"curve": curve, "through": "h2_msm x 16 + h2_ntt x 15 on pageable host columns (INTEGRATION.md section 3)",
            "ms_per_step": round(dt * 1e3, 3), "field_ops_per_s": ops_step / dt,
            "h2_msm_2e%d_ms" % k: round(t_msm * 1e3, 3), "h2_ntt_2e%d_ms" % (k + ext): round(t_big * 1e3, 3),
            "bytes_over_pcie_per_step": 16 * n * 32 + 16 * 96 + 2 * 32 * (7 * n + 8 * (n << ext)),
            "note": "every call uploads its column, runs one launch sequence and downloads the result; the resident step "
                    "(value) keeps columns in HBM and batches the columns of a phase into one launch"}


def headline_msm(args, lg, make_srs, world, rank, dev, stream, barrier, modmul, p, multi=False):
    """BASELINE config 4 / the north star's named target: ONE MSM of 2^20 terms.  N = 1: one launch sequence on one
    GPU.  N > 1: contiguous point-range split, rank r runs bases [n r / N, n (r+1) / N) against the replicated table,
    the N partial sums are all-gathered (96 B each) and added on the device.  Timed with the barrier + max-over-ranks
    bracket of the contract; the split result is checked against the rank's own unsplit MSM."""
    import torch
    import torch.distributed as dist
    import halo2_prover_amd as h2
    from halo2_prover_amd import sharded
    L = h2.load()
    cid = h2.CURVES[args.curve]
    n = 1 << lg
    bases = make_srs(n, 0x2468ACE)
    col = _device_scalars(L, cid, n, 0x5A, dev, stream)
    reps = 5

    def run(mode):
        return sharded.msm_phase_device(bases, col.data_ptr(), n, 1, stream, mode=mode)

    multi = multi or world > 1
    mode = "range" if multi else "single"
    run(mode)
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = run(mode)
    barrier()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ok = None
    if multi:
        q = BASE_FIELD[args.curve]
        whole = run("single")
        torch.cuda.synchronize()
        ok = jac_to_affine_ints(out.cpu().numpy().view(np.uint64), q) == \
            jac_to_affine_ints(whole.cpu().numpy().view(np.uint64), q)
        if not ok:
            print("rank %d: RANGE-SPLIT MSM DIFFERS from the unsplit one (split_equals_unsplit: false in the line)" % rank,
                  file=sys.stderr)
    ms = dt / reps * 1e3
    plan = bases.plan()
    gbs = (n * 96 + 96) / (ms * 1e-3) / 1e9
    rec = {"workload": "one MSM of 2^%d terms, %s" % (lg, args.curve), "n_gpus": world, "ms": round(ms, 4),
           "field_ops_per_s": ops_msm(n) / (ms * 1e-3), "algorithmic_GBps": round(gbs, 2),
           "hbm_frac": round(gbs / HBM_PEAK_GBS, 5), "window_bits": plan["window_bits"], "windows": plan["windows"],
           "split": "contiguous point ranges, all-gather of N x 96 B, device add" if world > 1 else "none",
           "split_equals_unsplit": ok}
    if modmul:
        mm = n * plan["windows"] * 10 / world / (ms * 1e-3)
        rec["modmul_per_s_per_gpu"] = mm
        rec["modmul_frac"] = round(mm / modmul["at_8_waves_per_simd"], 4)
        rec["modmul_frac_executed"] = round(rec["modmul_frac"] * PRODUCT_EQUIVALENTS_PER_ADD / 10.0, 4)
    bases.release()
    del col
    torch.cuda.empty_cache()
    return rec


def config5(args, make_srs, world, rank, dev, stream, barrier, p, gen, two_adicity, R):
    """BASELINE config 5: a 2^24-row domain with 64 advice columns, column j -> rank j mod N (64 / N whole columns per
    GPU): 64 MSMs against one replicated SRS + 64 NTTs, one all-gather of the 64 commitments.  Scalars are made on the
    device (8 x 512 MiB per GPU at N = 8)."""
    import torch
    import torch.distributed as dist
    import halo2_prover_amd as h2
    from halo2_prover_amd import sharded
    L = h2.load()
    cid = h2.CURVES[args.curve]
    lg = args.config5_k
    rehearsal = dist.get_backend() != "nccl" and lg > 18
    if rehearsal:
        lg = 18                 # ranks sharing one GPU over gloo: a rehearsal of the code path, not config 5's size
    n = 1 << lg
    total_cols = 64
    mine = len(range(rank, total_cols, world))
    bases = make_srs(n, 0x13579BD)
    cols = _device_scalars(L, cid, mine * n, 0xC5, dev, stream)
    out = torch.zeros((mine, 12), dtype=torch.int64, device=dev)
    root = pow(gen, (p - 1) >> two_adicity, p)
    w = limbs(pow(root, 1 << (two_adicity - lg), p) * R % p)

    def run():
        bases.msm_device(cols.data_ptr(), n, mine, out.data_ptr(), stream)
        allr = sharded.all_gather_rows(out)                       # the commitment vector, 64 x 96 B
        h2.ntt_device(cols.data_ptr(), mine, w, lg, args.curve, stream)
        return allr

    run()
    barrier()
    t0 = time.perf_counter()
    run()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ops = total_cols * (ops_msm(n) + ops_ntt(n))
    rec = {"workload": "2^%d rows x %d columns: %d MSM + %d NTT, column j -> rank j mod N" % (lg, total_cols, total_cols, total_cols),
           "n_gpus": world, "columns_per_gpu": mine, "ms": round(dt * 1e3, 2), "field_ops_per_s": ops / dt,
           "algorithmic_GBps": round(total_cols * n * (96 + 64) / dt / 1e9, 2), "rehearsal_size": rehearsal}
    bases.release()
    del cols
    torch.cuda.empty_cache()
    return rec


class _RecordedStream:
    """the deterministic RNG stream under which the reference's proofs were recorded:
    call i yields the leading bytes of SHA256("seed0-" + str(i)) digests"""

    def __init__(self):
        self.counter = 0

    def fill(self, nbytes):
        import hashlib
        out = b""
        while len(out) < nbytes:
            out += hashlib.sha256(b"seed0-%d" % self.counter).digest()
            self.counter += 1
        return out[:nbytes]

    def fr_random(self, _field=None):
        v = 0
        for i in range(8):
            v |= int.from_bytes(self.fill(8), "little") << (64 * i)
        return v % MODULI["bn254_fr"]


# sha256 of the proofs the reference's own build produced for Poseidon([1, 2])
REFERENCE_PROOF_SHA256 = {6: "6d235bf4637e1dce12559c44eaf77812bae2746d78331db3850e16b26234e63e",
                          11: "8d2d9052b47d9c9b45f3e3c268cec30797f74990cb47367bdfa7fbe77832129c",
                          16: "4c4e7d9301b652969a92718b3183f0bda79be2aaab245b68033ca96bf27bdc3c"}


def proof_generation(k):
    """proof-gen ms of the metric: the reference's wasm_generate_proof path (ParamsKZG::read + keygen + create_proof,
    KZG/GWC over BN254) for the Poseidon circuit at 2^k rows, through the C ABI's product surface (h2_setup /
    h2_generate_proof / h2_verify_proof: C++ orchestration, every column resident in HBM), under the recorded RNG stream
    so that the proof can be compared with the reference's own (bit-identical <=> equal sha256).  The Python mirror
    (prover.py, same bytes) is timed beside it."""
    import hashlib
    import torch
    import halo2_prover_amd as h2
    from halo2_prover_amd import lib as h2lib
    from halo2_prover_amd import prover
    L = h2.load()
    rng = _RecordedStream()

    def fill(_ctx, out, n):
        data = rng.fill(n)
        for i in range(n):
            out[i] = data[i]
    cb = h2lib.RNG_FILL(fill)
    js = ('{"x":[1,2],"output":"0x%064x"}' % prover.PoseidonCircuit([1, 2]).output()).encode()
    cap = 4 + 128 * (1 << k) + 256
    pbuf = ctypes.create_string_buffer(cap)
    ln = ctypes.c_size_t(0)
    t0 = time.perf_counter()
    h2lib.check(L.h2_setup(k, cb, None, pbuf, cap, ctypes.byref(ln)), "h2_setup")
    t1 = time.perf_counter()
    params = pbuf.raw[:ln.value]
    after_setup = rng.counter
    out = ctypes.create_string_buffer(1 << 16)
    runs = []
    # 0: cold (modules load, params parsed, arenas grow); 1, 2: steady, keys rebuilt on every call as wasm.rs does;
    # 3: params re-read too; 4, 5, 6: the library's default -- SRS tables and proving key kept between calls
    L.h2_key_cache(0)
    for i in range(7):
        rng.counter = after_setup            # the same draws again: every run must give the recorded proof
        if i == 3:
            L.h2_params_cache_clear()
        if i == 4:
            L.h2_key_cache(1)
        ta = time.perf_counter()
        h2lib.check(L.h2_generate_proof(params, len(params), js, 2, cb, None, out, 1 << 16, ctypes.byref(ln)), "h2_generate_proof")
        tb = time.perf_counter()
        runs.append((tb - ta, hashlib.sha256(out.raw[:ln.value]).hexdigest(), ln.value))
    proof = out.raw[:ln.value]
    ok = ctypes.c_int(0)
    tv = time.perf_counter()
    h2lib.check(L.h2_verify_proof(params, len(params), proof, len(proof), js, 2, ctypes.byref(ok)), "h2_verify_proof")
    verify_ms = (time.perf_counter() - tv) * 1e3
    digest, nbytes = runs[1][1], runs[1][2]
    same = (all(r[1] == REFERENCE_PROOF_SHA256[k] for r in runs)) if k in REFERENCE_PROOF_SHA256 else None
    # the same calls with the library's own randomness (getrandom) instead of the recorded stream's Python callback --
    # what a host that does not replay a recording pays; the last of these proofs goes through the verifier
    os_rng = {}
    for cache, key in ((0, "proof_gen_os_rng_ms"), (1, "create_proof_os_rng_ms")):
        L.h2_key_cache(cache)
        best = 1e9
        for i in range(4):
            ta = time.perf_counter()
            h2lib.check(L.h2_generate_proof(params, len(params), js, 2, None, None, out, 1 << 16, ctypes.byref(ln)), "h2_generate_proof")
            if i:
                best = min(best, time.perf_counter() - ta)
        os_rng[key] = round(best * 1e3, 2)
    ok2 = ctypes.c_int(0)
    h2lib.check(L.h2_verify_proof(params, len(params), out.raw[:ln.value], ln.value, js, 2, ctypes.byref(ok2)), "h2_verify_proof")
    os_rng["os_rng_proof_verified"] = bool(ok2.value)
    # the Python mirror on the same stream (keygen + create_proof, params already parsed).  Not inside a multi-rank
    # group: the mirror shards its commit phases over the RANKS (collectives), and only rank 0 is here
    import torch.distributed as dist
    alone = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
    rng.counter = after_setup
    pparams = prover.ParamsKZG.read(params) if alone else None
    mirror = []
    for _ in range(2 if alone else 0):
        rng.counter = after_setup
        ta = time.perf_counter()
        circuit = prover.PoseidonCircuit([1, 2])
        pk = prover.generate_keys(pparams, circuit)
        pproof = prover.generate_proof_with_instance(pparams, pk, circuit, [circuit.output()], rng)
        torch.cuda.synchronize()
        mirror.append((time.perf_counter() - ta, hashlib.sha256(pproof).hexdigest()))
        del pk
    del pparams
    return {"circuit": "poseidon (bn254, KZG/GWC)", "k": k, "n_gpus": int(L.h2_device_count()), "through": "C ABI: h2_generate_proof",
            "setup_ms": round((t1 - t0) * 1e3, 1),
            "proof_gen_ms": round(min(runs[1][0], runs[2][0]) * 1e3, 2),
            "create_proof_ms": round(min(runs[5][0], runs[6][0]) * 1e3, 2),
            "with_params_read_ms": round(runs[3][0] * 1e3, 2),
            "proof_gen_first_call_ms": round(runs[0][0] * 1e3, 1),
            "verify_ms": round(verify_ms, 1), "verified": bool(ok.value), **os_rng,
            "python_mirror_proof_gen_ms": round(mirror[1][0] * 1e3, 1) if mirror else None,
            "python_mirror_same_bytes": (mirror[1][1] == digest) if mirror else None,
            "proof_bytes": nbytes, "proof_sha256": digest, "bit_identical_to_reference": same,
            "note": "proof_gen_ms = one h2_generate_proof call with the key rebuilt as wasm_generate_proof does: JSON, keygen "
                    "on the empty circuit, witness, create_proof; wall clock, SRS tables resident from an earlier call "
                    "(best of two steady calls); create_proof_ms = the same call with the library's default key cache "
                    "(keygen excluded: SURVEY 8(d) asks for both); "
                    "with_params_read_ms = the same call after h2_params_cache_clear, i.e. with ParamsKZG::read as "
                    "wasm_generate_proof does on every call (parse, H2D, both MSM tables rebuilt); "
                    "proof_gen_first_call_ms = first call in the process; verify_ms = h2_verify_proof (key kept, "
                    "transcript, 2 small MSMs, host pairing); *_os_rng_ms = proof_gen_ms / create_proof_ms with the "
                    "library's getrandom instead of the recorded stream served by a Python callback (best of three)"}


def cpu_baseline(args, cols_np, n, k, ext, p, gen, two_adicity, R):
    """Time the CPU oracle (restatement of best_multiexp / best_fft, oracle/h2_oracle.c) on this host with all
    its cores.  Bounded sample of the same workload: for the proof-shaped step, 8 MSM(2^k) + 4 NTT(2^k) +
    4 NTT(2^(k+3)) -- half a step, the step's 16 : 7 : 8 mix -- about 10-30 core-seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    cid = O.CURVE_IDS[args.curve]
    fid = O.CURVE_SCALAR_FIELD[cid]
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = max(1, cores)                              # T = hardware_concurrency(); T = 64 and T = 1 beside it
    if args.workload == "ntt":
        n_m = 0
    elif args.workload == "poseidon":
        n_m = 8
    else:
        n_m = 2
    bases = O.synth_bases(cid, 0x48324D53000000B5, n, threads=threads) if n_m else None
    root = pow(gen, (p - 1) >> two_adicity, p)

    def om(lg):
        return limbs(pow(root, 1 << (two_adicity - lg), p) * R % p)

    t0 = time.perf_counter()
    ncols = max(1, cols_np.shape[0] // n)
    for j in range(n_m):
        jj = j % ncols
        O.best_multiexp(cid, cols_np[jj * n:(jj + 1) * n], bases, threads=threads)
    t_msm = time.perf_counter() - t0
    work = n_m * ops_msm(n)
    t_ntt = 0.0
    if args.workload != "msm":
        a = cols_np[:n].copy()
        big = splitmix_columns(99, n << ext, p) if args.workload == "poseidon" else None
        reps = 4 if args.workload == "poseidon" else 1
        t1 = time.perf_counter()
        for _ in range(reps):
            O.best_fft(fid, a, om(k), k, threads=threads)
            work += ops_ntt(n)
            if big is not None:
                O.best_fft(fid, big, om(k + ext), k + ext, threads=threads)
                work += ops_ntt(n << ext)
        t_ntt = time.perf_counter() - t1
    total = t_msm + t_ntt
    # the same port on ONE thread (BASELINE.md section 3 asks for both): one MSM and one NTT of each size
    single = None
    if args.workload == "poseidon":
        t2 = time.perf_counter()
        O.best_multiexp(cid, cols_np[:n], bases, threads=1)
        a1 = cols_np[:n].copy()
        O.best_fft(fid, a1, om(k), k, threads=1)
        O.best_fft(fid, big, om(k + ext), k + ext, threads=1)
        t_single = time.perf_counter() - t2
        single = {"value": (ops_msm(n) + ops_ntt(n) + ops_ntt(n << ext)) / t_single, "unit": "field-ops/s", "cores": 1,
                  "sample": "1 MSM(2^%d) + 1 NTT(2^%d) + 1 NTT(2^%d): %.2f s" % (k, k, k + ext, t_single)}
    at_64 = None
    if args.workload == "poseidon" and threads > 64:
        t3 = time.perf_counter()
        for j in range(4):
            O.best_multiexp(cid, cols_np[(j % ncols) * n:((j % ncols) + 1) * n], bases, threads=64)
        a64 = cols_np[:n].copy()
        O.best_fft(fid, a64, om(k), k, threads=64)
        O.best_fft(fid, big, om(k + ext), k + ext, threads=64)
        t64 = time.perf_counter() - t3
        at_64 = {"value": (4 * ops_msm(n) + ops_ntt(n) + ops_ntt(n << ext)) / t64, "unit": "field-ops/s", "cores": 64,
                 "sample": "4 MSM(2^%d) + 1 NTT(2^%d) + 1 NTT(2^%d): %.2f s" % (k, k, k + ext, t64)}
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": work / total, "unit": "field-ops/s", "cores": threads, "kind": "port", "cpu_model": model,
            "host_cores": cores, "single_thread": single, "at_64_threads": at_64,
            "sample": "%d MSM(2^%d)%s%s with %d threads: %.2f s" %
                      (n_m, k, " + %d NTT(2^%d)" % (4 if args.workload == "poseidon" else 1, k) if args.workload != "msm" else "",
                       " + 4 NTT(2^%d)" % (k + ext) if args.workload == "poseidon" else "", threads, total),
            "msm_s_per_call": (t_msm / n_m) if n_m else None}


if __name__ == "__main__":
    main()
