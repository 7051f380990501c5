"""event-timed NTT launches: python tools/time_ntt.py LOG_N COLS [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import halo2_prover_amd as h2
lg, m = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
h2.init(0)
p = bench.MODULI["pasta_fq"]
R = (1 << 256) % p
root = pow(5, (p - 1) >> 32, p)
w = bench.limbs(pow(root, 1 << (32 - lg), p) * R % p)
a = torch.from_numpy(bench.splitmix_columns(7, m << lg, p).view(np.int64)).cuda()
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    h2.ntt_device(a.data_ptr(), m, w, lg, "pallas", st)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    h2.ntt_device(a.data_ptr(), m, w, lg, "pallas", st)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
print("ntt 2^%d x %d: median %.1f us  min %.1f us" % (lg, m, ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
