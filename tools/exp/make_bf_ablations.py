"""Round 6: generated, git-ignored ablation copies of the F(5,4) bf16 x 6 kernel (results wrong, timing valid): what a K step's
ingredients cost.   python tools/exp/make_bf_ablations.py && tools/exp/bf_ablate.sh
  noa      the weights' pieces are requested once per segment (no loads in the K loop)
  nobar    no block barrier per chunk (the slab is still staged: racy)
  nosplit  the transformed values go to the MFMAs as raw bits (no split into three pieces)
  noxf     no transform (sample 0's values are split)
  nolds    one tap group's samples are read once per chunk and column ... (kp 0's reads serve all four channel pairs)
Each copy lives in tools/exp/abl_<name>/ (conv_wino54_kernel.h + conv_wino54_bf.hip)."""
from pathlib import Path

root = Path(__file__).resolve().parents[2]
hdr = (root / "flowhigh_amd/csrc/conv_wino54_kernel.h").read_text()
tu = (root / "flowhigh_amd/csrc/conv_wino54_bf.hip").read_text()


def emit(name, text):
    assert text != hdr, name
    d = root / "tools/exp" / f"abl_{name}"
    d.mkdir(exist_ok=True)
    (d / "conv_wino54_kernel.h").write_text(text)
    (d / "conv_wino54_bf.hip").write_text(tu)


emit("noa", hdr.replace("          load_a3(S, same_chunk ? c : c + 1, same_chunk ? g + 1 : 0, same_chunk || has_next);\n", ""))
emit("nobar", hdr.replace("      if (has_next) {\n        store_x(xbuf ^ 1);\n        __syncthreads();\n        xbuf ^= 1;",
                          "      if (has_next) {\n        store_x(xbuf ^ 1);\n        if (!BF) __syncthreads();\n        xbuf ^= 1;"))
emit("nosplit", hdr.replace("            v_split8(v, bh, bm, bl);\n",
                            "            bh = __builtin_bit_cast(v_bf16x8, (f32x4){v[0], v[1], v[2], v[3]}); bm = __builtin_bit_cast(v_bf16x8, (f32x4){v[4], v[5], v[6], v[7]});\n"
                            "            bl = __builtin_bit_cast(v_bf16x8, (f32x4){v[0] + v[4], v[1], v[2], v[7]});\n"))
emit("noxf", hdr.replace("                float t = __builtin_fmaf(bco[0][0], x[0][e], x[5][e]);\n#pragma unroll\n"
                         "                for (int j = 1; j < 5; ++j) t = __builtin_fmaf(bco[j][0], x[j][e], t);\n",
                         "                float t = x[0][e] + x[5][e];\n"))
emit("nolds", hdr.replace("              for (int j = 0; j < 6; ++j) x[j] = *reinterpret_cast<const f32x2*>(xsb + toff[g][j] + kp * V_PAIR + 64 * nt);\n",
                          "              for (int j = 0; j < 6; ++j) x[j] = *reinterpret_cast<const f32x2*>(xsb + toff[g][j] + 64 * nt);\n"))
print("written")
