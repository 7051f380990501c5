#!/bin/bash
# usage (on the GPU box): bash tools/ptimeline.sh  -- kernel timeline of the LAST of 3 key-cached Poseidon k=16 proofs: gaps and kernels
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
rm -rf gpurun_out/ptl
H2_PROFILE_KEY_CACHE=1 rocprofv3 --kernel-trace -d gpurun_out/ptl -o run --output-format csv -- python3 tools/proof_profile.py 3 > gpurun_out/ptl_out.txt 2> gpurun_out/ptl_err.txt
f=$(find gpurun_out/ptl -name '*kernel_trace.csv' | head -1)
python3 - "$f" > gpurun_out/ptimeline.txt <<'P'
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last proof: everything after the last long idle gap (> 300 us) preceding an expr_kernel... simpler: split at gaps > 400 us
groups, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - max(x[1] for x in cur) > 400000:
        groups.append(cur); cur = []
    cur.append(b)
groups.append(cur)
g = [x for x in groups if any('expr_kernel' in k[2] for k in x)][-1]
t0 = g[0][0]; end = t0; busy = 0
for s, e, name in g:
    gap = s - end
    busy += max(0, e - max(s, end))
    nm = name.replace('void ', '').split('(')[0][:46]
    print("%8.1f %8.1f %s%s" % ((s - t0) / 1000, (e - s) / 1000, nm.ljust(48), ("   <-- idle %.0f us before" % (gap / 1000)) if gap > 15000 else ""))
    end = max(end, e)
print("span %.1f us, GPU busy (union) %.1f us, kernels %d" % ((end - t0) / 1000, busy / 1000, len(g)))
P
rm -rf gpurun_out/ptl
tail -1 gpurun_out/ptimeline.txt
