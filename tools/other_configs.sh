#!/bin/bash
# Round 6: the other workloads at HEAD on one GPU (profiles/r06_other_configs.txt): the five BASELINE rows at their per-GPU sizes,
# short clips, and a 10 s clip under the odd-upsampler configuration at full width (ODD_CFG x 1536).  Default conv form, then
# conv_form=winograd for the BASELINE rows.
cd "$(dirname "$0")/.."
for form in "" winograd; do
  echo "== conv_form=${form:-default (bf16x6)}"
  export FH_CONV_FORM=$form; [ -z "$form" ] && unset FH_CONV_FORM
  echo "# configs[0] one 2 s clip 12->48 kHz euler x 1";              python tools/config_bench.py 1 2 12000 euler 1 20 2>&1 | grep "real time"
  echo "# configs[1] B=1 10 s 12->48 kHz euler x 1";                  python tools/config_bench.py 1 10 12000 euler 1 20 2>&1 | grep "real time"
  echo "# configs[2] B=32 10 s 16->48 kHz midpoint x 1";              python tools/config_bench.py 32 10 16000 midpoint 1 4 2>&1 | grep "real time"
  echo "# configs[3] one GPU's share: B=32 10 s 8->48 kHz euler x 1"; python tools/config_bench.py 32 10 8000 euler 1 4 2>&1 | grep "real time"
  echo "# configs[4] B=8 30 s 24->48 kHz midpoint x 4";               python tools/config_bench.py 8 30 24000 midpoint 4 3 2>&1 | grep "real time"
done
unset FH_CONV_FORM
echo "== short clips (default form): 0.5 / 1 / 5 s"
for s in 0.5 1 5; do python tools/config_bench.py 1 $s 12000 euler 1 20 2>&1 | grep "real time"; done
echo "== odd-upsampler configuration at full width"
python tools/odd_cfg_bench.py 2>&1 | grep "real time"
