"""h2_generate_proof / h2_verify_proof timings of the reference's three circuits at k = 16 through the C ABI
(key rebuilt on every call as wasm.rs does, then with the key kept); one JSON line."""
import ctypes, json, os, sys, time
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
seq = [25, 76, 38, 19, 58, 29, 88, 44, 22, 11, 34, 17, 52, 26, 13, 40, 20, 10, 5, 16, 8, 4, 2, 1]
jobs = [("collatz (SHPLONK)", 0, ('{"x":%s}' % str(seq).replace(" ", "")).encode()),
        ("arithmetic (GWC)", 1, b'{"x":6,"y":9,"constant":7,"z":2923}'),
        ("poseidon (GWC)", 2, ('{"x":[1,2],"output":"0x%064x"}' % prover.PoseidonCircuit([1, 2]).output()).encode())]
out = ctypes.create_string_buffer(1 << 16)
res = {"k": k, "through": "C ABI", "circuits": {}}
for name, idx, js in jobs:
    r = {}
    for cache in (0, 1):
        L.h2_key_cache(cache)
        best = 1e9
        for i in range(4):
            t = time.perf_counter()
            h2lib.check(L.h2_generate_proof(params, len(params), js, idx, None, None, out, 1 << 16, ctypes.byref(ln)), "prove")
            dt = (time.perf_counter() - t) * 1e3
            if i:
                best = min(best, dt)
        r["proof_gen_ms" if cache == 0 else "create_proof_ms"] = round(best, 2)
    proof = out.raw[:ln.value]
    ok = ctypes.c_int(0)
    best = 1e9
    for i in range(3):
        t = time.perf_counter()
        h2lib.check(L.h2_verify_proof(params, len(params), proof, len(proof), js, idx, ctypes.byref(ok)), "verify")
        best = min(best, (time.perf_counter() - t) * 1e3)
    r["verify_ms"] = round(best, 2)
    r["verified"] = bool(ok.value)
    r["proof_bytes"] = len(proof)
    res["circuits"][name] = r
print(json.dumps(res))
