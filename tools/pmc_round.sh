#!/bin/bash
# Run on the GPU box (gpurun): three rocprofv3 --pmc passes over one bench step, per-kernel averages -> gpurun_out/pmc_<tag>.txt
tag=${1:-r01}
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
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/p$i --output-format csv -- python3 bench.py ${PROF_ARGS:-} --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $out/p$i.log 2>&1
  python3 tools/pmc_generic.py $out/p$i > $out/p$i.txt
done
cat $out/p1.txt $out/p2.txt $out/p3.txt > gpurun_out/pmc_$tag.txt
rm -rf $out/p*/
wc -l gpurun_out/pmc_$tag.txt
