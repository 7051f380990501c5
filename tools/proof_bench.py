import sys, json
sys.path.insert(0, "/root/repo")
import bench, torch
import halo2_prover_amd as h2
h2.init(0)
print(json.dumps(bench.proof_generation(16)))
