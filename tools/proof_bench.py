"""proof-gen timings of bench.proof_generation(k) in a process of its own: python tools/proof_bench.py [k] [devices]
`devices` > 1: h2_init_devices([0, ..., devices - 1]) -- the C++ prover spreads its commit phases over the contexts;
a comma list (e.g. 0,0) names the device ids explicitly (two contexts on one GPU: a rehearsal of the code path)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import halo2_prover_amd as h2
from halo2_prover_amd import api

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
spec = sys.argv[2] if len(sys.argv) > 2 else "1"
ids = [int(x) for x in spec.split(",")] if "," in spec else list(range(int(spec)))
if len(ids) > 1:
    api.init_devices(ids)
else:
    h2.init(ids[0])
rec = bench.proof_generation(k)
rec["device_ids"] = ids
print(json.dumps(rec))
