"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<name>.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o run --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o run --output-format csv -- python3 bench.py ...
    python3 tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/pmc_traffic.json [step_executions]

The summary is stamped with a hash of the kernel sources (bench.py's source_hash) and H2_GIT_HEAD from the
environment: bench.py reports `traffic` from it only while the sources are the ones that were profiled.
`step_executions` = how many times the profiled command ran the step's NTTs (warmup + steps + the 3 runs of the
phases_ms measurement), to turn the NTT launches' total into bytes per step.

Correction (MI355X_MICROARCH.md, HBM section; checked in round 1 on the then msm_table_kernel, whose only read was 4 MiB of bases): on gfx950
FETCH_SIZE counts half the bytes of 16-byte-per-lane loads, which is what every kernel here issues, so
traffic = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KB)."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d + "/run_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            acc[name].append(float(r["Counter_Value"]))
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    step_execs = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    import bench
    f, w = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(f) | set(w)):
        fk = sum(f[name]) / len(f[name]) if f.get(name) else 0.0
        wk = sum(w[name]) / len(w[name]) if w.get(name) else 0.0
        kernels[name] = {"FETCH_SIZE_KB_avg_per_launch": round(fk, 1), "launches": len(f.get(name) or w.get(name)),
                         "WRITE_SIZE_KB_avg_per_launch": round(wk, 1),
                         "traffic_bytes_per_launch": int((2 * fk + wk) * 1024)}
    dom = next(k for k in kernels if "msm_chunk_kernel" in k)
    doc = {"source_hash": bench.source_hash(), "git_head": os.environ.get("H2_GIT_HEAD", "?"), "steps_profiled": step_execs,
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) -- python3 "
                      "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-proof --no-extras",
           "workload": "poseidon_k16_proof_shape, pallas", "units": "KB as reported by rocprofv3",
           "correction": "traffic = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 halves FETCH_SIZE for 16-B-per-lane loads; "
                         "calibrated in round 1 on the then msm_table_kernel: n * 64 B = 4096 KB of bases read)",
           "dominant_kernel": dict(kernels[dom], name=dom), "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print(dom, kernels[dom])


if __name__ == "__main__":
    main()
