#!/bin/bash
# Build timing-experiment variants of the Winograd kernel next to the real library and time them with
#   FH_LIB_PATH=flowhigh_amd/lib/abl/abl<N>.so python tools/wino_time.py 768 5000 1
# WINO_ABL bits: 1 = LDS reads stay but no B^T transform, 4 = A tile of the first step only (no weight
# loads in the loop), 8 = slab of the first chunk only (no slab loads / stores / barriers).  Results are wrong
# by construction; only the time matters.  The activation kernel has the same for ACT_ABL (1 = no sin^2,
# 2 = no up filter / snake, 4 = no down filter): tools/act_ablate.sh.
set -e
cd "$(dirname "$0")/.."
mkdir -p flowhigh_amd/lib/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DWINO_ABL=$n -c flowhigh_amd/csrc/conv_wino.hip -o /tmp/conv_wino_abl$n.o
  objs=$(ls flowhigh_amd/build/*.o | grep -v conv_wino)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowhigh_amd/lib/abl/abl$n.so $objs /tmp/conv_wino_abl$n.o
done
