#!/bin/bash
# Where does the time of the 24-channel residual-stack launch (direct kernel, 32 x 512 tile) go?  Ablation builds of
# flowhigh_amd/csrc/conv_mfma.hip (experiment patches applied to a copy; nothing here touches the product source):
#   noepi : no residual loads, no output stores (one lane writes one element so that the work is not optimised away)
#   nomfma: the K loop without its MFMAs (loads, LDS staging, barriers and fragment reads stay)
# Run on the GPU box: tools/exp/conv24_ablate.sh > gpurun_out/conv24_ablate.txt
set -e
cd "$(dirname "$0")/../.."
# (the ablations were measured on the experiment build with the 24 channels as one chunk; the patches below apply to the product source as well)
python - <<'PY'
s = open("flowhigh_amd/csrc/conv_mfma.hip").read()
a = s.replace("          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q][nt] * scale), ro, off[q][nt], 0, 0);",
              "          if (v[q][nt] * scale == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q][nt]), ro, off[q][nt], 0, 0);")
a = a.replace("      if (nres > 0) {\n        float t0[4][NT];", "      if (nres > 99) {\n        float t0[4][NT];")
open("tools/exp/_conv_noepi.hip", "w").write(a)
b = s.replace("""          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks >> 2][mt][ks & 3], bf[ks & 1][nt],
                                                             acc[mt][nt], 0, 0, 0);""",
              """          acc[mt][nt][ks & 15] += a[ks >> 2][mt][ks & 3] * bf[ks & 1][nt];""")
open("tools/exp/_conv_nomfma.hip", "w").write(b)
PY
bash tools/build_variant.sh c_noepi tools/exp/_conv_noepi.hip=conv_mfma.hip > /dev/null 2>&1
bash tools/build_variant.sh c_nomfma tools/exp/_conv_nomfma.hip=conv_mfma.hip > /dev/null 2>&1
cat > /tmp/c24.py <<'PY'
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import conv_bench as cb
for res in (False, True):
    cb.run(24, 480000, [11, 7, 3], 4, res=res, label="stage5 stack res=%d" % res)
    cb.run(24, 480000, [11, 11, 11], 4, res=res, label="  k=11 x3 res=%d" % res)
    cb.run(24, 480000, [3, 3, 3], 4, res=res, label="  k=3 x3 res=%d" % res)
PY
for v in "" c_noepi c_nomfma; do
  echo "== ${v:-product}"
  if [ -z "$v" ]; then python /tmp/c24.py 2>&1 | grep -v amdgpu; else FH_LIB_PATH=tools/abl/$v.so python /tmp/c24.py 2>&1 | grep -v amdgpu; fi
done
