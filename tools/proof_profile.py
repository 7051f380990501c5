"""N proofs of Poseidon k = 16 through the C ABI with the key cache off (keygen every call, as wasm.rs does):
run under `rocprofv3 --kernel-trace --stats` to see where a proof's GPU time goes (totals / N)."""
import ctypes
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_prover_amd as h2
from halo2_prover_amd import lib as h2lib, prover

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
h2.init(0)
L = h2.load()
cap = 4 + 128 * (1 << k) + 256
pbuf = ctypes.create_string_buffer(cap)
ln = ctypes.c_size_t(0)
h2lib.check(L.h2_setup(k, None, None, pbuf, cap, ctypes.byref(ln)), "h2_setup")
params = pbuf.raw[:ln.value]
js = ('{"x":[1,2],"output":"0x%064x"}' % prover.PoseidonCircuit([1, 2]).output()).encode()
out = ctypes.create_string_buffer(1 << 16)
L.h2_key_cache(int(os.environ.get('H2_PROFILE_KEY_CACHE', '0')))   # 0: keygen every call (as wasm.rs does)
import time
for i in range(N + 1):
    t = time.perf_counter()
    h2lib.check(L.h2_generate_proof(params, len(params), js, 2, None, None, out, 1 << 16, ctypes.byref(ln)), "prove")
    if i == N:
        print("last proof %.2f ms" % ((time.perf_counter() - t) * 1e3))
