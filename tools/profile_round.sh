#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + the two HBM-traffic PMC passes + the bench line.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ with tools/make_profile_summary.py.
# PROF_ARGS: extra bench.py arguments of EVERY pass (e.g. PROF_ARGS="--batch 32" for BASELINE configs[2]'s batch).
tag=${1:-r02}
suf=${2:-B1}        # file-name suffix of the workload: B1 (default), or e.g. B32 with PROF_ARGS="--batch 32"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
# the activation occupancy cap is measured once here and fixed for the profiled runs (otherwise the ~640 calibration
# launches of every process would sit in the kernel statistics)
blocks=$(python3 -c "
import sys; sys.path.insert(0, '.')
from flowhigh_amd import vocoder as V
print(V.calibrate_act_occupancy('cuda:0', bf=V.use_bf16x6()))" 2> gpurun_out/calibrate_$tag.err | tail -1)
case "$blocks" in
  0|2|3|4|5) export FH_ACT_BLOCKS=$blocks ;;
  *) echo "calibration failed (got '$blocks', see gpurun_out/calibrate_$tag.err): every run calibrates by itself" >&2; unset FH_ACT_BLOCKS ;;
esac
echo "FH_ACT_BLOCKS=${FH_ACT_BLOCKS:-auto}"
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py ${PROF_ARGS:-} --steps 16 --warmup 4 --no-cpu-baseline --no-alt > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 bench.py ${PROF_ARGS:-} --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 bench.py ${PROF_ARGS:-} --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $out/pmc_write.log 2>&1
python tools/make_traffic_json.py $out/pmc_fetch $out/pmc_write $out/conv_hbm_bytes_per_launch.json conv > /dev/null
python tools/make_traffic_json.py $out/pmc_fetch $out/pmc_write $out/act_hbm_bytes_per_launch.json act > /dev/null
if [ "$suf" = B1 ]; then cp $out/conv_hbm_bytes_per_launch.json $out/act_hbm_bytes_per_launch.json profiles/
else for k in conv act; do cp $out/${k}_hbm_bytes_per_launch.json profiles/${k}_hbm_bytes_per_launch_$suf.json; done; fi
python bench.py ${PROF_ARGS:-} ${BENCH_ARGS:-} 2> $out/bench.err | tail -1 > $out/bench_line.json
cp $(ls $out/stats/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/stats/*/*_kernel_trace.csv $out/pmc_fetch/*/*agent* $out/pmc_write/*/*agent*
tail -1 $out/stats.log | cut -c1-200; cat $out/bench_line.json | cut -c1-900
