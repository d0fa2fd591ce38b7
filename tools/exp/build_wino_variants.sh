#!/bin/bash
# Experiment build of the library with the two rejected Winograd block shapes of round 3 compiled in as tiles 8 / 9:
#   tools/exp/conv_wino2.hip  4-wave blocks, all six transform points of an output in one wave (A^T in registers)
#   tools/exp/conv_wino3.hip  persistent 12-wave workgroups of two independent 6-wave teams
# -> flowhigh_amd/lib/abl/winox.so.  Use: FH_LIB_PATH=flowhigh_amd/lib/abl/winox.so python tools/wino2_bench.py
# (both kernels are bit-identical to the product's tiles and slower: profiles/r03_wino_block_shapes.txt)
set -e
cd "$(dirname "$0")/../.."
python -m flowhigh_amd.build > /dev/null
mkdir -p flowhigh_amd/lib/abl
H="/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iflowhigh_amd/csrc -Itools/exp -DFH_WINO_EXPERIMENTS"
$H -c flowhigh_amd/csrc/conv_wino.hip -o /tmp/winox_conv_wino.o
$H -c tools/exp/conv_wino2.hip -o /tmp/winox_conv_wino2.o
$H -c tools/exp/conv_wino3.hip -o /tmp/winox_conv_wino3.o
objs=$(ls flowhigh_amd/build/*.hip.o | grep -v "/conv_wino.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowhigh_amd/lib/abl/winox.so $objs /tmp/winox_conv_wino.o /tmp/winox_conv_wino2.o /tmp/winox_conv_wino3.o
echo flowhigh_amd/lib/abl/winox.so
