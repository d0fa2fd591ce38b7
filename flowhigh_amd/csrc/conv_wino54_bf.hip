// bf16 x 6 instantiations of the F(5,4) wide-stage conv (conv_wino54_kernel.h, `BF`): a translation unit of their own because
// they are compiled with -fno-slp-vectorize (flowhigh_amd/build.py): at plain -O3 the SLP vectoriser re-packs the scalar FMAs /
// subtractions of the transform and the split into v_pk_fma_f32 / v_pk_add_f32, which stall the bf16 MFMAs they sit beside.
// The fp32-MFMA instantiations (conv_wino54.hip) keep the default flags and their code.
#include "conv_wino54_kernel.h"

int fh_internal_wino54_bf(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation, int pm, int mt,
                          bool vl, hipStream_t st, const int* run_map, int n_runs) {
#define FH_W54BF_CASE(MT)                                                                                                       \
  case MT:                                                                                                                      \
    return vl ? launch_wino54<MT, true, false, true>(groups, n_groups, batch, cout_pad, len, dilation, pm, st, run_map, n_runs)  \
              : launch_wino54<MT, false, false, true>(groups, n_groups, batch, cout_pad, len, dilation, pm, st, run_map, n_runs);
  switch (mt) {          // (a 128-row block fits with 2 spilled registers and is no faster per row: 3.00 us per K step against
                         // 2.22 us of the 96-row block -- the loop is not bound by the vector work an extra row tile amortises)
    FH_W54BF_CASE(3)
    FH_W54BF_CASE(2)
  }
#undef FH_W54BF_CASE
  fh_set_error("fh_conv_wino54_f32: no bf16 x 6 block of %d row tiles", mt);
  return FH_E_ARG;
}
