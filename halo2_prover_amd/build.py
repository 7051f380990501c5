"""Build libh2hip.so in-tree with hipcc for gfx950 (one translation unit: csrc/h2_capi.hip)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "h2_capi.hip")
OUT = os.path.join(HERE, "libh2hip.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in
        ("h2_capi.hip", "h2_msm.hpp", "h2_ntt.hpp", "h2_curve.hpp", "h2_field.hpp", "h2_constants.inc")]
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "h2hip.h"))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
