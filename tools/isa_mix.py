"""Instruction mix per kernel of a hipcc -S listing: python tools/isa_mix.py file.s [name filter]
(packed-fp32 / plain VALU / MFMA / LDS / scratch counts: what sits beside the matrix instructions)"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = re.split(r'\n(?=_Z\w+:)', s)
pat = re.compile(r'^\s+(v_pk_\w+|v_mfma\w+|v_cvt_pk_bf16_f32|v_fma_f32|v_fmac_f32\w*|v_sub_f32\w*|v_add_f32\w*|v_mul_f32\w*|v_and_b32\w*|v_lshlrev_b32\w*'
                 r'|v_perm_b32|ds_read\w*|ds_write\w*|scratch_\w+|buffer_load\w+|s_waitcnt|s_nop|s_barrier|v_accvgpr\w+|v_mov_b32\w*)', re.M)
for f in funcs:
    m = re.match(r'(_Z\w+):', f)
    if not m or flt not in m.group(1):
        continue
    body = f.split('s_endpgm')[0]
    c = collections.Counter(pat.findall(body))
    tot_v = len(re.findall(r'^\s+v_', body, re.M))
    tot_s = len(re.findall(r'^\s+s_', body, re.M))
    print(m.group(1)[:110])
    print("   total v_", tot_v, " s_", tot_s, " ", {k: v for k, v in sorted(c.items())})
