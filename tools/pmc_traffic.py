"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<name>.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o run --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o run --output-format csv -- python3 bench.py ...
    python3 tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/pmc_traffic.json [step_executions]

The summary is stamped with a hash of the kernel sources (bench.py's source_hash) and H2_GIT_HEAD from the
environment: bench.py reports `traffic` from it only while the sources are the ones that were profiled.
`step_executions` = how many times the profiled command ran the step's NTTs (warmup + steps + the 3 runs of the
phases_ms measurement), to turn the NTT launches' total into bytes per step.

Correction PER ACCESS SHAPE (profiles/pmc_calibration.json, measured with tools/pmc_calib.hip on known byte counts;
MI355X_MICROARCH.md calibrates the streaming shape only and asks for exactly this): on gfx950 FETCH_SIZE counts ONE
64-byte unit per memory request, and a request is up to 128 bytes:
    16 B per lane streaming, contiguous runs >= 128 B     requested = 2.00 x FETCH_SIZE
    one 64-byte row per lane at random rows (gather)       requested = 0.95-1.01 x FETCH_SIZE
    64-byte runs with a large stride                       requested = 0.99 x FETCH_SIZE
WRITE_SIZE is exact for 16-B-per-lane streams and 64-byte runs; a scattered 4-byte store is written as 32 bytes.
Kernels by shape: msm_chunk_kernel = gather (its table rows; round 2 doubled it: 2.2x what the kernel can request);
ntt29_pass_kernel = 64-byte runs for 1024-row tiles (2 columns, workgroups of 512), >= 128-byte runs otherwise; every
other kernel reads streams (scalars, sorted entries, 144-byte points)."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(d, counter):
    """kernel name -> list of (value in KB, workgroup size) per dispatch"""
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d + "/run_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            acc[name].append((float(r["Counter_Value"]), int(r.get("Workgroup_Size", 0) or 0)))
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    step_execs = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    import bench
    cal = json.load(open(os.path.join(ROOT, "profiles", "pmc_calibration.json")))["shapes"]
    f_stream = cal["stream_read_16B_per_lane"]["factor"]
    f_gather = cal["gather_64B_rows_table_64MiB"]["factor"]      # the 2^16 tables (80 MiB) sit in the Infinity Cache
    f_run64, f_run128 = cal["strided_runs_64B"]["factor"], cal["strided_runs_128B"]["factor"]
    f, w = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(f) | set(w)):
        fl, wl = f.get(name, []), w.get(name, [])
        if "msm_chunk_kernel" in name:
            shape, read_kb = "gather_64B_rows", [v * f_gather for v, _ in fl]
        elif "ntt29_pass_kernel" in name:
            shape = "64B runs (workgroups of 512: 1024-row tiles) or >= 128B runs"
            read_kb = [v * (f_run64 if wg == 512 else f_run128) for v, wg in fl]
        else:
            shape, read_kb = "stream", [v * f_stream for v, _ in fl]
        fk = sum(v for v, _ in fl) / len(fl) if fl else 0.0
        rk = sum(read_kb) / len(read_kb) if read_kb else 0.0
        wk = sum(v for v, _ in wl) / len(wl) if wl else 0.0
        kernels[name] = {"FETCH_SIZE_KB_avg_per_launch": round(fk, 1), "launches": len(fl or wl),
                         "WRITE_SIZE_KB_avg_per_launch": round(wk, 1), "read_shape": shape,
                         "read_KB_avg_per_launch_calibrated": round(rk, 1),
                         "traffic_bytes_per_launch": int((rk + wk) * 1024)}
    dom = next(k for k in kernels if "msm_chunk_kernel" in k)
    doc = {"source_hash": bench.source_hash(), "git_head": os.environ.get("H2_GIT_HEAD", "?"), "steps_profiled": step_execs,
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) -- python3 "
                      "bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-proof --no-extras",
           "workload": "poseidon_k16_proof_shape, pallas", "units": "KB as reported by rocprofv3",
           "correction": "per access shape, factors from profiles/pmc_calibration.json (tools/pmc_calib.hip): stream x%.2f, "
                         "64-byte row gather x%.2f, 64-byte runs x%.2f, >= 128-byte runs x%.2f; WRITE_SIZE as reported"
                         % (f_stream, f_gather, f_run64, f_run128),
           "dominant_kernel": dict(kernels[dom], name=dom), "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print(dom, kernels[dom])


if __name__ == "__main__":
    main()
