#!/bin/bash
# Round 6 stage A: the bf16 x 6 K loop of the F(4,3) kernel with plain (one result per lane) fp32 VALU against the packed form
# of round 5 (tools/abl/prev.so = HEAD~, tools/abl/bfplain.so = this tree with -fno-slp-vectorize).  K-step constants per tile.
cd "$(dirname "$0")/../.."
for lib in tools/abl/prev.so tools/abl/bfplain.so; do
  echo "== $lib"
  FH_LIB_PATH=$lib python tools/wino_cost_fit.py
  for shape in "768 5000 1" "384 20000 1" "192 60000 3" "96 120000 5"; do
    FH_LIB_PATH=$lib python tools/wino_time.py $shape bf
  done
done
