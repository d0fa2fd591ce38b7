#!/bin/bash
# Build timing-experiment variants of the Winograd kernel (WINO_ABL bit mask) next to the real library.
set -e
cd "$(dirname "$0")/.."
mkdir -p flowhigh_amd/lib/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DWINO_ABL=$n -c flowhigh_amd/csrc/conv_wino.hip -o /tmp/conv_wino_abl$n.o
  objs=$(ls flowhigh_amd/build/*.o | grep -v conv_wino)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowhigh_amd/lib/abl/abl$n.so $objs /tmp/conv_wino_abl$n.o
done
