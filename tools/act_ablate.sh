#!/bin/bash
# Timing-experiment variants of the activation kernel (ACT_ABL bits: 1 = no sin^2, 2 = no up filter / snake,
# 4 = no down filter; 6 = data movement only).  Time with
#   FH_LIB_PATH=flowhigh_amd/lib/abl/actabl<N>.so python tools/act_bench.py
set -e
cd "$(dirname "$0")/.."
mkdir -p flowhigh_amd/lib/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DACT_ABL=$n -c flowhigh_amd/csrc/act1d.hip -o /tmp/act_abl$n.o
  objs=$(ls flowhigh_amd/build/*.o | grep -v act1d)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowhigh_amd/lib/abl/actabl$n.so $objs /tmp/act_abl$n.o
done
