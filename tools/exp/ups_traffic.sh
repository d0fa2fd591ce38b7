#!/bin/bash
# HBM bytes written / read by the direct-kernel upsampler launches, phase-fused (default) against one group per phase
# (FH_UPS_FUSE=0): rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE over one bench step.  GPU box: bash tools/exp/ups_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for f in 1 0; do
  for c in WRITE_SIZE FETCH_SIZE; do
    out=gpurun_out/ups_traffic_${f}_$c; rm -rf $out
    FH_UPS_FUSE=$f FH_ACT_BLOCKS=0 rocprofv3 --pmc $c -d $out --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt > /dev/null 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for f in (1, 0):
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for ci, c in enumerate(("WRITE_SIZE", "FETCH_SIZE")):
        fn = glob.glob(f"gpurun_out/ups_traffic_{f}_{c}/**/*_counter_collection.csv", recursive=True)[0]
        for r in csv.DictReader(open(fn)):
            if "conv_mfma_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("conv_mfma_kernel")[1].split("(")[0]
                agg[k][ci] += float(r["Counter_Value"])
                agg[k][2] += ci == 0
    print(f"FH_UPS_FUSE={f}: per kernel over 2 steps: launches, MB written, MB read (FETCH_SIZE x 2 per the guide)")
    for k, (w, rd, n) in sorted(agg.items()):
        print(f"  conv_mfma_kernel{k:24s} {n:3d}  written {w * 1024 / 1e6:8.1f} MB  read {2 * rd * 1024 / 1e6:8.1f} MB")
PY
rm -rf gpurun_out/ups_traffic_*
