#!/bin/bash
# Round 6: the bench step with the bf16 x 6 convs under each activation occupancy cap (does the cap that protects the fp32-MFMA
# conv launches' clock still pay when the conv launches run on the bf16 pipe?)
cd "$(dirname "$0")/../.."
for bf in 0 1; do for cap in 0 3 4; do
  FH_CONV_BF16X6=$bf FH_ACT_BLOCKS=$cap python bench.py --steps 40 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('bf16x6=$bf act cap $cap:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms  conv', d['roofline']['conv_ms_per_step'], 'act', d['roofline_hbm']['act_ms_per_step'])"
done; done
