#!/usr/bin/env python3
"""Extract the reference's own known-answer DATA for Pasta field arithmetic into JSON.

Reads (as text, in the build container only -- /root/reference does not exist on the
GPU box) the byte arrays of the 44 zcash Orchard Poseidon vectors held by the
reference's tests:

  /root/reference/circuits/src/poseidon/primitives/test_vectors.rs:16   fp permute (11)
  /root/reference/circuits/src/poseidon/primitives/test_vectors.rs:420  fp hash    (11)
  /root/reference/circuits/src/poseidon/primitives/test_vectors.rs:641  fq permute (11)
  /root/reference/circuits/src/poseidon/primitives/test_vectors.rs:1045 fq hash    (11)

plus three spot values of the constants tables (first round constant, MDS[0][0]) from
fp.rs:13,1307 / fq.rs:13,1307 that pin the Grain/MDS generation.

Output: tests/golden/pasta_poseidon_vectors.json -- inputs and expected outputs only
(32-byte little-endian canonical field elements as hex strings).  No source text of the
reference is kept.
"""
import json
import os
import re
import sys

REF = "/root/reference/circuits/src/poseidon/primitives"


def byte_arrays(text):
    """all [0x.., .. ] 32-byte arrays in order."""
    out = []
    for m in re.finditer(r"\[\s*((?:0x[0-9a-fA-F]{2},\s*){31}0x[0-9a-fA-F]{2},?)\s*\]", text):
        bs = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", m.group(1)))
        assert len(bs) == 32
        out.append(bs.hex())
    return out


def first_from_raw(text, after):
    """first `from_raw([a,b,c,d])` after marker -> canonical int."""
    i = text.index(after)
    m = re.search(r"from_raw\(\[\s*(0x[0-9a-f_]+),\s*(0x[0-9a-f_]+),\s*(0x[0-9a-f_]+),\s*(0x[0-9a-f_]+),?\s*\]\)",
                  text[i:])
    limbs = [int(x.replace("_", ""), 16) for x in m.groups()]
    return sum(l << (64 * j) for j, l in enumerate(limbs))


def main():
    tv = open(os.path.join(REF, "test_vectors.rs")).read()
    fp_mod, fq_mod = tv.index("pub(crate) mod fp"), tv.index("pub(crate) mod fq")
    out = {}
    for name, seg in (("fp", tv[fp_mod:fq_mod]), ("fq", tv[fq_mod:])):
        h = seg.index("pub(crate) fn hash()")
        perm = byte_arrays(seg[:h])
        hsh = byte_arrays(seg[h:])
        assert len(perm) == 11 * 6 and len(hsh) == 11 * 3, (len(perm), len(hsh))
        out[name] = {
            "permute": [{"initial_state": perm[6 * i:6 * i + 3], "final_state": perm[6 * i + 3:6 * i + 6]}
                        for i in range(11)],
            "hash": [{"input": hsh[3 * i:3 * i + 2], "output": hsh[3 * i + 2]} for i in range(11)],
        }
        ct = open(os.path.join(REF, name + ".rs")).read()
        out[name]["round_constant_0_0"] = "%064x" % first_from_raw(ct, "ROUND_CONSTANTS")
        out[name]["mds_0_0"] = "%064x" % first_from_raw(ct, "const MDS:")
        out[name]["mds_inv_0_0"] = "%064x" % first_from_raw(ct, "const MDS_INV:")
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pasta_poseidon_vectors.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    sys.exit(main())
