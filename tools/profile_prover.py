"""cProfile of the GPU prover's host side at Poseidon k=16 (run on the GPU box)."""
import cProfile
import hashlib
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import halo2_prover_amd as h2  # noqa: E402
from halo2_prover_amd import prover  # noqa: E402
from bench import _RecordedStream  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
h2.init(0)
rng = _RecordedStream()
params = prover.generate_params(k, rng)
circuit = prover.PoseidonCircuit([1, 2])
t0 = time.perf_counter()
prk = cProfile.Profile()
prk.enable()
pk = prover.generate_keys(params, circuit)
torch.cuda.synchronize()
prk.disable()
t1 = time.perf_counter()
ctr = rng.counter
prover.generate_proof_with_instance(params, pk, circuit, [circuit.output()], rng)    # warm-up (modules, arenas)
rng.counter = ctr
torch.cuda.synchronize()
t1 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
proof = prover.generate_proof_with_instance(params, pk, circuit, [circuit.output()], rng)
torch.cuda.synchronize()
pr.disable()
t2 = time.perf_counter()
print("keygen %.3f s, create_proof %.3f s, sha256 %s" % (t1 - t0, t2 - t1, hashlib.sha256(proof).hexdigest()[:16]))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
print("---- keygen ----")
pstats.Stats(prk).sort_stats("cumulative").print_stats(28)
