"""One Poseidon k = 16 proof, then three h2_verify_proof calls with the phase trace (run with H2_TRACE=1)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_prover_amd as h2
from halo2_prover_amd import lib as h2lib, prover
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
h2.init(0)
L = h2.load()
cap = 4 + 128 * (1 << k) + 256
pbuf = ctypes.create_string_buffer(cap)
ln = ctypes.c_size_t(0)
h2lib.check(L.h2_setup(k, None, None, pbuf, cap, ctypes.byref(ln)), "h2_setup")
params = pbuf.raw[:ln.value]
js = ('{"x":[1,2],"output":"0x%064x"}' % prover.PoseidonCircuit([1, 2]).output()).encode()
out = ctypes.create_string_buffer(1 << 16)
h2lib.check(L.h2_generate_proof(params, len(params), js, 2, None, None, out, 1 << 16, ctypes.byref(ln)), "prove")
proof = out.raw[:ln.value]
ok = ctypes.c_int(0)
for i in range(3):
    t = time.perf_counter()
    h2lib.check(L.h2_verify_proof(params, len(params), proof, len(proof), js, 2, ctypes.byref(ok)), "verify")
    print("verify %d: ok=%d %.2f ms" % (i, ok.value, (time.perf_counter() - t) * 1e3), flush=True)
