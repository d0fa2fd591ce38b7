#!/bin/bash
# The boxes of the pool differ by up to 10 % on the batch-1 headline (495-547 audio-s/s).  One line per resource, to see which
# one differs on THIS box: register-only MFMA rate and clock, device copy bandwidth, one Winograd launch (sustained), the conv
# launches of a step, the activation launches, the bench value.   bash tools/box_probe.sh > gpurun_out/box_probe.txt
cd "$(dirname "$0")/.."
echo "== $(hostname) $(date -u +%H:%M:%S)"
tools/micro/mfma_mix 2>/dev/null | sed -n 1p
python tools/copy_bench.py 2>/dev/null | tail -2
python tools/wino_sustained.py 2>/dev/null | tail -3
if [ -f tools/abl/trace2.so ]; then       # in-kernel shader clock of a conv launch, alone and after activation launches
  for alt in 0 1; do echo -n "ALT=$alt: "; ALT=$alt WARM=200 FH_LIB_PATH=tools/abl/trace2.so python tools/wino_trace2.py 192 60000 1 0 1 2>/dev/null | grep "launch\|shader clock" | tr '\n' ' '; echo; done
fi
python tools/conv_layers.py 1 1000 2>/dev/null | tail -1
python tools/act_bench.py 2>/dev/null | sed -n 3,4p
python bench.py --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('bench',d['value'],'audio-s/s',d['ms_per_step'],'ms  conv frac',d['roofline']['frac'],' act frac',d['roofline_hbm']['frac'])"
