"""Timing-experiment (ablation) builds of the product kernels.  The product sources carry no experiment switches: this
script applies textual patches to COPIES of them (tools/exp/_abl_<name>.hip), builds each as a variant library
(tools/build_variant.sh -> tools/abl/<name>.so) and prints how to time it.  Results of ablation builds are
wrong by construction; only the time matters.

    python tools/exp/ablations.py wino_notransform wino_noweights wino_noslab act_nosin act_noup act_nodown act_dataonly
    FH_LIB_PATH=tools/abl/<name>.so python tools/wino_time.py 768 5000 1      (or tools/act_bench.py)
"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]

# name -> (source file, [(old snippet, new snippet), ...])
WINO_T = ('''          asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(bf[nt]) : "s"(c0), "v"(xr[p & 1][0][nt]), "v"(xr[p & 1][1][nt]));
          asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(qq) : "s"(c1), "v"(xr[p & 1][2][nt]), "v"(xr[p & 1][3][nt]));
          if (nt + 1 < NT)
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(bf[nt]) : "s"(c2), "v"(qq));
          else
            asm("v_pk_fma_f32 %0, %1, %2, %0\\n\\ts_nop 1" : "+v"(bf[nt]) : "s"(c2), "v"(qq));''',
          '''          bf[nt] = xr[p & 1][0][nt];      // ablation: the LDS reads stay, no B^T transform
          asm volatile("" : "+v"(bf[nt]) : "v"(xr[p & 1][1][nt]), "v"(xr[p & 1][2][nt]), "v"(xr[p & 1][3][nt]), "v"(qq));''')
WINO_A = ('''        if (kp & 1) {                              // this half of the A registers is free: refill it for the next step''',
          '''        if (false) {                               // ablation: A tile of the first step only (no weight loads in the loop)''')
WINO_S1 = ('''      auto stage = [&](int p) {
''', '''      auto stage = [&](int p) {
        return;                                    // ablation: slab of the first chunk only (no loads / stores / barriers)
''')
WINO_S2 = ('''      if constexpr (VL) {
        if (has_slab) {
          if (GC == 1) vl_store(Ss, xbuf ^ 1, sub, 0, 0);''', '''      if constexpr (false) {
        if (has_slab) {
          if (GC == 1) vl_store(Ss, xbuf ^ 1, sub, 0, 0);''')
WINO_S3 = ('''      if (has_next && sub == SUBS - 1) {                 // end of a slab: one barrier per SUBS chunks''',
           '''      if (false) {''')
WINO_E = ('''      if (eact) {
        const float* er = E + (eth * 6 * 32 + ecol) * W_EP + 4 * erq;''', '''      if (eact && tid == 12345) {                 // ablation: no epilogue reads / arithmetic / stores (the exchange writes stay)
        const float* er = E + (eth * 6 * 32 + ecol) * W_EP + 4 * erq;''')
ACT_SIN = ('''        s2[r] = sin_squared2(arg[r]);
        amax = fmaxf(fmaxf(amax, fabsf(arg[r][0])), fabsf(arg[r][1]));     // (v_max3; NaN falls through, as sinf(NaN))''',
           '''        s2[r] = arg[r];                            // ablation: no sin^2''')
ACT_UP = ('''      for (int r = 0; r < ACT_PPT; ++r) zout[r] = __builtin_elementwise_fma((f32x2)(inv_beta), s2[r], zf[r]);
      f32x4* zw''', '''      for (int r = 0; r < ACT_PPT; ++r) zout[r] = (f32x2){xv[r + 1], xv[r + 2]} * alpha + inv_beta;   // ablation: no up filter / snake
      f32x4* zw''')
ACT_DOWN = ('''        f32x2 a2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 6; ++j)
          a2 = __builtin_elementwise_fma((f32x2){zv[2 * r + 2 + 2 * j], zv[2 * r + 3 + 2 * j]}, fdp[j], a2);
        // (one scalar add per output: left to the vectoriser this became 6 moves + 2 packed adds per tile)
        asm("v_add_f32 %0, %1, %2" : "=v"(out[r]) : "v"(a2[0]), "v"(a2[1]));
      }
      // outputs whose taps leave''', '''        out[r] = zv[2 * r + 8];                    // ablation: no down filter
      }
      // outputs whose taps leave''')
def act_occ(dyn_bytes):      # unused dynamic LDS per block: limits the resident blocks per CU (12.7 KB static + this, of 160 KB)
    return ("dim3(256), 0, (hipStream_t)stream", f"dim3(256), {dyn_bytes}, (hipStream_t)stream")


RECIPES = {
    "wino_notransform": ("conv_wino.hip", [WINO_T]),
    "wino_noweights": ("conv_wino.hip", [WINO_A]),
    "wino_noslab": ("conv_wino.hip", [WINO_S1, WINO_S2, WINO_S3]),
    "wino_noslab_noweights": ("conv_wino.hip", [WINO_A, WINO_S1, WINO_S2, WINO_S3]),
    "wino_mfma_lds_only": ("conv_wino.hip", [WINO_T, WINO_A, WINO_S1, WINO_S2, WINO_S3]),
    "wino_noepilogue": ("conv_wino.hip", [WINO_E]),
    "act_nosin": ("act1d.hip", [ACT_SIN]),
    "act_noup": ("act1d.hip", [ACT_UP]),
    "act_nodown": ("act1d.hip", [ACT_DOWN]),
    "act_dataonly": ("act1d.hip", [ACT_UP, ACT_DOWN]),
    "act_occ5": ("act1d.hip", [act_occ(18 * 1024)]),     # 5 blocks (20 waves) per CU instead of 7
    "act_occ4": ("act1d.hip", [act_occ(26 * 1024)]),
    "act_occ3": ("act1d.hip", [act_occ(38 * 1024)]),
    "act_occ2": ("act1d.hip", [act_occ(50 * 1024)]),
}


def build(name):
    src, patches = RECIPES[name]
    s = (ROOT / "flowhigh_amd" / "csrc" / src).read_text()
    for old, new in patches:
        if s.count(old) < 1:
            raise SystemExit(f"{name}: a patch anchor no longer matches {src}:\n{old}")
        s = s.replace(old, new)
    tmp = ROOT / "tools" / "exp" / f"_abl_{name}.hip"
    tmp.write_text(s)
    try:
        subprocess.check_call(["bash", str(ROOT / "tools" / "build_variant.sh"), name, f"tools/exp/_abl_{name}.hip={src}"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    finally:
        tmp.unlink()
    print(f"tools/abl/{name}.so")


if __name__ == "__main__":
    names = sys.argv[1:] or list(RECIPES)
    for n in names:
        build(n)
