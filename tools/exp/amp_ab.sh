for lib in "" $(ls tools/abl/*.so 2>/dev/null); do
  echo "== ${lib:-product}"
  for a in "48 240000 1" "48 240000 3" "24 480000 1" "24 480000 5"; do FH_LIB_PATH=${lib:-flowhigh_amd/lib/libflowhigh_hip.so} python tools/amp_bench.py $a 2>/dev/null | grep "act=False"; done
done
