#!/bin/bash
# Round 6: K-step constants of the F(5,4) bf16 x 6 kernel with one ingredient removed at a time (tools/exp/make_bf_ablations.py).
cd "$(dirname "$0")/../.."
for v in "" noa nobar nosplit noxf nolds; do
  lib=flowhigh_amd/lib/libflowhigh_hip.so
  [ -n "$v" ] && lib=tools/abl/bf_$v.so
  echo "== ${v:-product}"
  FH_LIB_PATH=$lib python tools/wino54_cost_fit.py bf 2>&1 | grep "^bf16x6"
done
