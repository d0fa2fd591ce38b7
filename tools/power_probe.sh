#!/bin/bash
# Board power and shader clock (rocm-smi, readable without root) while (a) the register-only MFMA loop with all the K loop's
# ingredients (tools/micro/mfma_mix) and (b) a real Winograd launch back to back (tools/wino_sustained.py) run: is the
# 2.07-2.15 GHz of the real launch a POWER limit?   tools/power_probe.sh   (GPU box)
probe() {  # $1 = label, rest = command
  label=$1; shift
  ( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | egrep -i "power|sclk" | tr '\n' ' '; echo; sleep 0.05; done ) > /tmp/probe_$label.txt &
  pp=$!
  "$@" > /tmp/run_$label.txt 2>&1
  kill $pp 2>/dev/null; wait $pp 2>/dev/null
  echo "== $label"; grep -v amdgpu /tmp/run_$label.txt | tail -4
  python3 - "$label" <<'PY'
import re, sys
rows = open(f"/tmp/probe_{sys.argv[1]}.txt").read().splitlines()
pw = [float(m.group(1)) for r in rows for m in [re.search(r"Power \(W\): ([0-9.]+)", r)] if m]
ck = [float(m.group(1)) for r in rows for m in [re.search(r"sclk[^(]*\(([0-9.]+)Mhz\)", r)] if m]
if pw: print(f"   power samples {len(pw)}: max {max(pw):.0f} W, top-5 mean {sum(sorted(pw)[-5:]) / min(5, len(pw)):.0f} W")
if ck: print(f"   sclk samples {len(ck)}: min {min(ck):.0f} max {max(ck):.0f} MHz")
if not pw and rows: print("   (no power lines parsed) sample:", rows[0][:200])
PY
}
rocm-smi --showpower --showclocks 2>&1 | head -20
probe mix tools/micro/mfma_mix
probe conv python tools/wino_sustained.py 192 60000
