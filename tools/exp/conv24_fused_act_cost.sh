#!/bin/bash
# What would hosting the activation's ARITHMETIC in the slab loader of the 24-channel conv cost the conv?  (VERDICT r02 item 4:
# "measure, rather than estimate, one fused activation -> conv pair at C = 24".)  A faithful fusion needs the 12-tap up / down
# filters across the loader's lanes (an LDS round trip more per chunk); this build injects only the arithmetic: every staged
# input sample goes through 33 dependent-free packed-equivalent FMAs (the strip kernel's count per sample: 7 + 13 + 2 + 8 + misc)
# between its global load and its LDS store, i.e. a LOWER bound of what the fused conv would pay.  Against it stands the
# activation launch it would save: 58-65 us per launch of this stage (tools/act_bench.py, C = 48 / 24 rows).
#   tools/exp/conv24_fused_act_cost.sh      (on the GPU box)
set -e
cd "$(dirname "$0")/../.."
python - <<'PY'
s = open("flowhigh_amd/csrc/conv_mfma.hip").read()
old = """        if (j < XW) dst[j] = __uint_as_float(xreg[rr][i]);"""
new = """        if (j < XW) {
          float v = __uint_as_float(xreg[rr][i]), a0 = v, a1 = 0.5f * v, a2 = 0.25f * v;
#pragma unroll
          for (int t = 0; t < 11; ++t) {        // 33 FMAs in three independent chains
            a0 = fmaf(a0, 0.999f, v); a1 = fmaf(a1, 0.998f, v); a2 = fmaf(a2, 0.997f, v);
          }
          dst[j] = v + 1e-30f * (a0 + a1 + a2);
        }"""
assert s.count(old) == 1
open("tools/exp/_conv_actcost.hip", "w").write(s.replace(old, new))
PY
bash tools/build_variant.sh c_actcost tools/exp/_conv_actcost.hip=conv_mfma.hip > /dev/null 2>&1
rm tools/exp/_conv_actcost.hip
cat > /tmp/c24b.py <<'PY'
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
sys.argv = sys.argv[:1] + ['x']
import conv_bench as cb
for res in (False, True):
    cb.run(24, 480000, [11, 7, 3], 4, res=res, label="stage5 stack res=%d" % res)
    cb.run(24, 480000, [3, 3, 3], 4, res=res, label="  k=3 x3 res=%d" % res)
PY
echo "== product"; python /tmp/c24b.py 2>&1 | grep -v amdgpu
echo "== + 33 FMAs per staged sample"; FH_LIB_PATH=tools/abl/c_actcost.so python /tmp/c24b.py 2>&1 | grep -v amdgpu
