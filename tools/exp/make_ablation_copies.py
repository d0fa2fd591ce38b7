"""Generated, git-ignored ablation copies of product kernels for round 5's timing experiments (results wrong, timing valid):
    python tools/exp/make_ablation_copies.py
      tools/exp/abl_nofirst/    the F(5,4) kernel (fp32 form) without the request of a K segment's first slab
      tools/exp/abl_nofirst2/   ... and without its first two A-fragment halves       (profiles/r05_w54_first_slab_bound.txt)
      tools/exp/amp_fused_onepercu.hip     the narrow-stage kernel launched with 40 KB of unused LDS and half the grid: one block
                                           per CU (build with -DF_LDS_PAD=40960; profiles/r05_amp_ablation.txt item 10)
    tools/build_variant.sh <name> tools/exp/abl_<copy>/conv_wino54.hip=conv_wino54.hip -Iinclude   (round 6: the kernel lives in
    conv_wino54_kernel.h, so a copy is a directory with the patched header and the unchanged conv_wino54.hip beside it; the
    bf16 x 6 form's ablations: make_bf_ablations.py)
    tools/build_variant.sh <name> tools/exp/amp_fused_onepercu.hip=amp_fused.hip -Iinclude -DF_LDS_PAD=40960"""
from pathlib import Path

root = Path(__file__).resolve().parents[2]
src = (root / "flowhigh_amd/csrc/conv_wino54_kernel.h").read_text()
tu = (root / "flowhigh_amd/csrc/conv_wino54.hip").read_text()


def emit(name, text):
    d = root / "tools/exp" / f"abl_{name}"
    d.mkdir(exist_ok=True)
    (d / "conv_wino54_kernel.h").write_text(text)
    (d / "conv_wino54.hip").write_text(tu)


a = src.replace("    load_x(S, 0, true);\n    store_x(xbuf);\n    __syncthreads();\n    for (int c = 0; c < nch; ++c) {",
                "    load_x(S, 0, false);\n    store_x(xbuf);\n    __syncthreads();\n    for (int c = 0; c < nch; ++c) {")
assert a != src
emit("nofirst", a)
b = a.replace("      load_a_half(0, S, 0, 0, true);\n      load_a_half(1, S, 0, 0, true);",
              "      load_a_half(0, S, 0, 0, false);\n      load_a_half(1, S, 0, 0, false);")
assert b != a
emit("nofirst2", b)
src = (root / "flowhigh_amd/csrc/amp_fused.hip").read_text()
key = "  hipLaunchKernelGGL((amp_actconv_kernel<MA, VEC>), dim3((unsigned)grid), dim3(F_THREADS), bytes, stream, groups, tiles,"
c = src.replace(key, """#ifdef F_LDS_PAD
  hipFuncSetAttribute((const void*)amp_actconv_kernel<MA, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes + F_LDS_PAD);
  hipLaunchKernelGGL((amp_actconv_kernel<MA, VEC>), dim3((unsigned)(total_tiles < resident / 2 ? total_tiles : resident / 2)), dim3(F_THREADS),
                     bytes + F_LDS_PAD, stream, groups, tiles, channels, dilation, total_tiles, cmax);
  return FH_OK;
#endif
""" + key)
assert c != src
(root / "tools/exp/amp_fused_onepercu.hip").write_text(c)
print("written")
