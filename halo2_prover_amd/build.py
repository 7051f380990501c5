"""Build libh2hip.so in-tree with hipcc for gfx950.

Five translation units: csrc/h2_curve_impl.hip once per curve (-DH2_CURVE_ID=0/1/2; the kernels),
csrc/h2_capi.hip (host logic, the C ABI) and csrc/h2_prover.hip (the product surface: keygen / prove / verify).  They are compiled in parallel and linked into one shared
library that travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libh2hip.so")
OBJ = os.path.join(HERE, "build")
DEPS = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
    os.path.join(os.path.dirname(HERE), "include", "h2hip.h"),
    os.path.join(os.path.dirname(HERE), "include", "h2hip_selftest.h"),
]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return OUT
    if os.environ.get("H2_BUILD_TUNING"):          # the sweep knobs of csrc/h2_tune.hpp (tools/sweep_*.sh): not a product build
        extra_flags = tuple(extra_flags) + ("-DH2_TUNING",)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for cid, name in ((0, "bn254"), (1, "pallas"), (2, "vesta")):
        obj = os.path.join(OBJ, "curve_%s.o" % name)
        jobs.append(([hipcc] + FLAGS + list(extra_flags) + ["-DH2_CURVE_ID=%d" % cid, "-c",
                     os.path.join(CSRC, "h2_curve_impl.hip"), "-o", obj], obj))
    capi = os.path.join(OBJ, "capi.o")
    jobs.append(([hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, "h2_capi.hip"), "-o", capi], capi))
    prover = os.path.join(OBJ, "prover.o")
    jobs.append(([hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, "h2_prover.hip"), "-o", prover], prover))

    def run(job):
        if verbose:
            print(" ".join(job[0]), file=sys.stderr)
        subprocess.check_call(job[0])
        return job[1]

    with ThreadPoolExecutor(max_workers=5) as ex:
        objs = list(ex.map(run, jobs))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
