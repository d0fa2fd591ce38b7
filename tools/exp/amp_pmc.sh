#!/bin/bash
# PMC passes over the narrow-stage kernel alone (tools/amp_bench.py runs it ~120 times per form).  On the GPU box:
#   bash tools/exp/amp_pmc.sh "24 480000 1"  ->  gpurun_out/amp_pmc_<C>_<d>.txt
args=${1:-"24 480000 1"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
set -- $args
out=gpurun_out/amp_pmc_$1_$3; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/p$i --output-format csv -- python3 tools/amp_bench.py $args 10 > $out/p$i.log 2>&1
  python3 tools/pmc_generic.py $out/p$i amp_actconv > $out/p$i.txt
done
cat $out/p*.txt > $out.txt
rm -rf $out
cat $out.txt
