#!/bin/bash
# Build an experiment variant of ONE kernel source next to the real library:
#   tools/build_variant.sh <name> <source.hip> [-DFLAG ...]   ->  flowhigh_amd/lib/abl/<name>.so
# and run anything against it with FH_LIB_PATH=flowhigh_amd/lib/abl/<name>.so (same C ABI).
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p flowhigh_amd/lib/abl
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -c flowhigh_amd/csrc/$src -o /tmp/variant_$name.o
objs=$(ls flowhigh_amd/build/*.hip.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowhigh_amd/lib/abl/$name.so $objs /tmp/variant_$name.o
echo flowhigh_amd/lib/abl/$name.so
