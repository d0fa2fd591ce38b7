#!/bin/bash
# Same-box A/B of the activation occupancy cap (FH_ACT_BLOCKS): classify the box (does a conv launch slow down after an
# activation launch?), the (activation, conv) pair time per cap as vocoder.calibrate_act_occupancy measures it, then the
# bench value with each cap forced, then with the calibration's own choice.
cd "$(dirname "$0")/../.."
python tools/clock_dip_probe.py 2>/dev/null | sed -n 1,2p | cut -c1-120
python - <<'PY' 2>/dev/null
import sys; sys.path.insert(0, '.')
from flowhigh_amd import vocoder as V
for rnd in range(2):
    print("pair us:", "  ".join(f"cap {b}: {V.measure_act_conv_pair('cuda:0', b):6.1f}" for b in (0, 5, 4, 3, 2)))
PY
for v in 0 4 3 2 0 4 3 2 auto; do
  FH_ACT_BLOCKS=$v python bench.py --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('FH_ACT_BLOCKS=$v ->',d['config']['act_blocks_per_cu'],d['value'],'audio-s/s',d['ms_per_step'],'ms  conv frac',d['roofline']['frac'],' act frac',d['roofline_hbm']['frac'])"
done
