"""Tuning aid: per-launch durations of msm_fixup_kernel from rocprofv3 kernel traces taken with different
H2_FIXUP_LOG_G overrides (see the gpurun command in DESIGN.md section 5).  Usage: fixup_sweep.py DIR..."""
import csv, sys, collections
for d in sys.argv[1:]:
    rows = [r for r in csv.DictReader(open(d + "/run_kernel_trace.csv"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for name in ("msm_fixup", "msm_weight", "msm_chunk_kernel"):
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
        per = collections.defaultdict(list)
        for i, x in enumerate(dur):
            per[i % 5].append(x)
        print(d, name, " ".join("%.0f" % (sorted(v)[len(v) // 2]) for _, v in sorted(per.items())))
