#!/bin/bash
# Kernel sequence of ONE bench step (rocprofv3 --kernel-trace): which launches sit between which.  On the GPU box:
#   bash tools/exp/step_trace.sh  ->  gpurun_out/step_trace.txt  (kernel names of the last step, in start order, with durations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
out=gpurun_out/step_trace; rm -rf $out; mkdir -p $out
FH_ACT_BLOCKS=${FH_ACT_BLOCKS:-0} rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $out/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/step_trace/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
# last step: from the last resample kernel on
idx = max(i for i, r in enumerate(rows) if 'resample' in r['Kernel_Name'])
with open('gpurun_out/step_trace.txt', 'w') as o:
    prev_end = None
    for r in rows[idx:]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        o.write(f"{short(r['Kernel_Name']):62s} {(e - s) / 1e3:9.1f} us   gap {gap:7.1f} us\n")
        prev_end = e
PY
rm -rf $out
grep -c . gpurun_out/step_trace.txt; grep -n "copyBuffer" -B2 -A1 gpurun_out/step_trace.txt | head -120
