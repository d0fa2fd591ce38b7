#!/bin/bash
# Registers, spills and occupancy of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage), one line
# per kernel.  No GPU needed:  tools/kernel_resources.sh conv_wino54.hip [-DFLAG ...]
cd "$(dirname "$0")/.."
src=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iflowhigh_amd/csrc -Iinclude -Rpass-analysis=kernel-resource-usage "$@" \
    -c flowhigh_amd/csrc/$src -o /tmp/kernel_resources.o 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size" \
  | sed 's/\[-Rpass[^]]*\]//g; s/^[^ ]*: remark: *//; s/remark: *//; s/Function Name: //' \
  | awk '/^_Z|^[a-zA-Z_]+[^:]*$/ {if (line) print line; line=$0; next} {gsub(/^ +/, ""); line=line "  |  " $0} END {print line}' \
  | sed 's/ \[bytes\/lane\]//; s/ \[waves\/SIMD\]//; s/ \[bytes\/block\]//' | c++filt | cut -c1-260
