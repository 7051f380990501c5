"""Check the SAMPLE lines of tools/microbench_limb29 against big integers: r = a b 2^-261 mod p and r < 2p."""
import sys
P = {"pasta_fp": 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
     "pasta_fq": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
     "bn254_fq": 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
     "bn254_fr": 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001}
n = bad = 0
for line in sys.stdin:
    if not line.startswith("SAMPLE"):
        continue
    w = line.split()
    p = P[w[1]]
    v = [int(x, 16) for x in w[2:]]
    val = lambda limbs: sum(x << (29 * i) for i, x in enumerate(limbs))
    a, b, r = val(v[0:9]), val(v[9:18]), val(v[18:27])
    ok = (r - a * b * pow(2, -261, p)) % p == 0 and r < 2 * p and all(x < (1 << 29) for x in v[18:26])
    n += 1
    bad += not ok
print("limb29 samples: %d checked, %d bad" % (n, bad))
sys.exit(1 if bad or not n else 0)
