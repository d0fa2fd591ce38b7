#!/bin/bash
# Build an experiment variant of ONE kernel source next to the real library:
#   tools/build_variant.sh <name> <source.hip> [-DFLAG ...]          (a file of flowhigh_amd/csrc, other flags)
#   tools/build_variant.sh <name> tools/exp/<file>.hip=<source.hip>  (an experimental copy that replaces <source.hip>)
#   ->  tools/abl/<name>.so ; run anything against it with FH_LIB_PATH=tools/abl/<name>.so
# (same C ABI).  Experiment code lives in tools/exp/, never in the product sources.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p tools/abl
if [[ "$src" == *=* ]]; then
  file=${src%%=*}; src=${src##*=}
else
  file=flowhigh_amd/csrc/$src
fi
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iflowhigh_amd/csrc "$@" -c $file -o /tmp/variant_$name.o
objs=$(ls flowhigh_amd/build/*.hip.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abl/$name.so $objs /tmp/variant_$name.o
echo tools/abl/$name.so
