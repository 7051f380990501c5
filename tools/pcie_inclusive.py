"""Host-pointer entry points against the device-resident ones (the PCIe-inclusive rate is
reported beside `value`, never as it).  Run on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo2_prover_amd as h2
from bench import splitmix_columns, MODULI, limbs

h2.init(0)
curve, k = "pallas", 16
n = 1 << k
p = MODULI["pasta_fq"]
R = (1 << 256) % p
buf = torch.empty((n, 8), dtype=torch.int64, device="cuda")
h2.lib.check(h2.load().h2_srs_generate(h2.CURVES[curve], limbs(0x1234567 * R % p).ctypes.data, n, buf.data_ptr(), None), "srs")
torch.cuda.synchronize()
bases = h2.Bases.from_device(curve, buf.data_ptr(), n)
m = 4
cols = splitmix_columns(7, m * n, p).reshape(m, n, 4)
dev = torch.from_numpy(cols.view(np.int64)).cuda()
out = torch.zeros((m, 12), dtype=torch.int64, device="cuda")
for name, fn in (("h2_msm_device, 4 columns resident", lambda: (bases.msm_device(dev.data_ptr(), n, m, out.data_ptr()), torch.cuda.synchronize())),
                 ("h2_msm_batch, 4 host columns (H2D + D2H inside)", lambda: bases.msm_batch(list(cols))),
                 ("h2_msm, 1 host column", lambda: bases.msm(cols[0]))):
    fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print("%-50s %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
a = cols[0].copy()
w = limbs(pow(pow(5, (p - 1) >> 32, p), 1 << (32 - k), p) * R % p)
d = torch.from_numpy(a.view(np.int64)).cuda()
for name, fn in (("h2_ntt_device, 1 column resident", lambda: (h2.ntt_device(d.data_ptr(), 1, w, k, curve), torch.cuda.synchronize())),
                 ("h2_ntt, 1 host column (H2D + D2H inside)", lambda: h2.best_fft(a, w, k, curve))):
    fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print("%-50s %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
