#!/usr/bin/env python3
"""Mint params_k4.bin / params_k6.bin: ParamsKZG::<Bn256>::new(k).write() under the deterministic
RNG stream of SURVEY.md App. B.2, re-derived by the big-int reference (oracle/pyref.py) and accepted only
if the sha256 equals the value recorded from the reference's own build."""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import pyref as R  # noqa: E402

WANT = {4: "e410bf985e9327e7ea474d74907e4209b50678844fab11bc09bcd1e2a2ae1272",
        6: "3cd009bb91fe7f1d4c2cc54296263062e5a1d2359aacd68bfa1ce542b5a1169d"}
s = R.SurveyStream().fr_random(R.BN_FR)
for k, want in WANT.items():
    data, _, _ = R.params_kzg_bytes(k, s)
    assert hashlib.sha256(data).hexdigest() == want, k
    open(os.path.join(HERE, "params_k%d.bin" % k), "wb").write(data)
    print("params_k%d.bin" % k, len(data), "bytes, sha256 ok")
