#!/bin/bash
# narrow_bf.hip variants (tools/abl/nb_<name>.so) next to the product build: tools/exp/nb_ab.sh "<names>" [C] [L]
cd "$(dirname "$0")/../.."
for v in "" $1; do
  lib=flowhigh_amd/lib/libflowhigh_hip.so
  [ -n "$v" ] && lib=tools/abl/nb_$v.so
  echo "== ${v:-product}"
  FH_LIB_PATH=$lib python tools/narrow_bench.py ${2:-24} ${3:-480000} 2>&1 | grep "^ *("
done
