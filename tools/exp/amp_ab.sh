for lib in "" tools/abl/amp_noahead.so tools/abl/amp_early3.so tools/abl/amp_late.so tools/abl/amp_late_noahead.so; do
  echo "== ${lib:-product}"
  for a in "48 240000 1" "48 240000 3" "24 480000 1" "24 480000 5"; do FH_LIB_PATH=${lib:-flowhigh_amd/lib/libflowhigh_hip.so} python tools/amp_bench.py $a 2>/dev/null | grep fused; done
done
