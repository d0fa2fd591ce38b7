#!/bin/bash
# Same-box A/B of the activation kernel's occupancy cap (tools/exp/ablations.py act_occ*): first classify the box
# (does a conv launch slow down after an activation launch?), then the bench value with each library.
cd "$(dirname "$0")/../.."
python tools/clock_dip_probe.py 2>/dev/null | sed -n 1,2p
for v in base act_occ4 act_occ3 base act_occ4 act_occ3; do
  if [ $v = base ]; then unset FH_LIB_PATH; else export FH_LIB_PATH=flowhigh_amd/lib/abl/$v.so; fi
  python bench.py --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v',d['value'],'audio-s/s',d['ms_per_step'],'ms  conv frac',d['roofline']['frac'],' act frac',d['roofline_hbm']['frac'])"
done
