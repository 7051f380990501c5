#!/bin/bash
# usage (on the GPU box): bash tools/step_timeline.sh -- kernel timeline of the LAST bench step (start, duration, queue, kernel)
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
rm -rf gpurun_out/stl
rocprofv3 --kernel-trace -d gpurun_out/stl -o run --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-proof --no-extras > gpurun_out/stl_out.txt 2> gpurun_out/stl_err.txt
f=$(find gpurun_out/stl -name '*kernel_trace.csv' | head -1)
python3 - "$f" > gpurun_out/step_timeline.txt <<'P'
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steps of the timed region: split at msm_digits kernels of the advice phase; simply take a window: find the 8 step starts
# = every 4th msm_digits launch (4 launch sequences per step) among the first 8 steps, print step index 6 (a timed one)
dig = [i for i, r in enumerate(rows) if 'msm_digits_kernel' in r[2]]
start = dig[4 * 6]
end = dig[4 * 7]
t0 = rows[start][0]
for s, e, name, q in rows[start:end]:
    nm = name.replace('void ', '').replace('h2::', '').split('(')[0].split('<')[0][:28]
    print("%8.1f %7.1f  q%-3s %s" % ((s - t0) / 1000, (e - s) / 1000, q, nm))
print("step span %.1f us" % ((rows[end][0] - t0) / 1000))
P
rm -rf gpurun_out/stl
cat gpurun_out/step_timeline.txt
