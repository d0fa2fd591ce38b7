"""Generates tools/exp/conv_wino54_trace.hip (not committed): flowhigh_amd/csrc/conv_wino54.hip with wall-clock stamps
(s_memrealtime, 100 MHz) of thread 0 of every block at the phase boundaries, written to a buffer set through
fh_w54_set_trace().  Build: tools/build_variant.sh w54trace tools/exp/conv_wino54_trace.hip=conv_wino54.hip ; run: tools/exp/w54_trace.py"""
from pathlib import Path
root = Path(__file__).resolve().parents[2]
s = (root / "flowhigh_amd/csrc/conv_wino54.hip").read_text()


def sub(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


sub('#include "fh_common.h"', '#include "fh_common.h"\n__device__ unsigned long long* g_w54_trace = nullptr;\n'
    '#define TR(k) do { if (g_w54_trace && threadIdx.x == 0) g_w54_trace[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)\n'
    'extern "C" int fh_w54_set_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_w54_trace), &p, sizeof(p)); }')
sub("  const int tid = threadIdx.x;\n", "  const int tid = threadIdx.x;\n  TR(0);\n"
    "  if (g_w54_trace && threadIdx.x == 0) {\n    unsigned hw, xcc;\n"
    "    asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hw));\n"
    "    asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc));\n"
    "    g_w54_trace[(size_t)blockIdx.x * 16 + 11] = ((unsigned long long)xcc << 32) | hw;\n  }\n")
sub("    setup_seg(S);\n", "    setup_seg(S);\n    TR(1);\n")
sub("    store_x(xbuf);\n    __syncthreads();\n", "    TR(2);\n    store_x(xbuf);\n    __syncthreads();\n    TR(3);\n")
sub("  // ---- epilogue ---", "  TR(4);\n  // ---- epilogue ---")
sub("  if (AHEAD && VL && nres > 0) request_res(0, rpre[0]);\n", "  if (AHEAD && VL && nres > 0) request_res(0, rpre[0]);\n  TR(5);\n")
sub("    const bool wave_vec = VL && __builtin_amdgcn_ballot_w64(!vec) == 0ull;\n    __syncthreads();\n",
    "    const bool wave_vec = VL && __builtin_amdgcn_ballot_w64(!vec) == 0ull;\n    if (mt == 0) TR(6);\n    __syncthreads();\n    if (mt == 0) TR(7);\n")
sub("  // (keeps pf alive:", "  TR(9);\n  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  TR(10);\n  // (keeps pf alive:")
# end of round 0 .. 2: stamp 8 after the first round's stores are issued
sub("    // (no barrier here: E is not the slab", "    if (mt == 1) TR(8);\n    // (no barrier here: E is not the slab")
(root / "tools/exp/conv_wino54_trace.hip").write_text(s)
print("tools/exp/conv_wino54_trace.hip")
