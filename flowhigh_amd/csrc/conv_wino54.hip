// C ABI of the F(5,4) wide-stage conv (kernel: conv_wino54_kernel.h).  The fp32-MFMA instantiations live here; the bf16 x 6
// ones in conv_wino54_bf.hip (compiled with -fno-slp-vectorize: build.py).
#include "conv_wino54_kernel.h"

extern "C" int fh_wino54_tile_m(int tile_cfg) {
  if (tile_cfg & FH_WINO_BF16X6) {           // bf16 x 6: 96- and 64-row blocks
    const int id = tile_cfg & ~FH_WINO_BF16X6;
    tile_cfg = (id == 1 || id == 2) ? id : -1;
  }
  return tile_cfg == 0 ? 128 : tile_cfg == 1 ? 96 : tile_cfg == 2 ? 64 : tile_cfg == 3 ? 48 : -1;
}
extern "C" int fh_wino54_tile_n(void) { return V_OUT; }

// bf16 x 6 instantiations (conv_wino54_bf.hip): mt = 3 / 2 row tiles of 32
int fh_internal_wino54_bf(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation, int pm, int mt,
                          bool vl, hipStream_t st, const int* run_map, int n_runs);

namespace {
int wino54_dispatch(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation,
                    int layout_flags, int tile_cfg, hipStream_t st, const int* run_map, int n_runs) {
  const int pm = layout_flags & 1;
  // 16-byte accesses need contiguous aligned rows (layout bit 1: the caller rules them out -- ragged launches in which
  // some group's rows are not 16-byte aligned; `len` is then only the longest group's length)
  // (phase-major rows are always aligned: bit 1 is about plain rows)
  const bool vl = pm || (dilation == 1 && len % 4 == 0 && !(layout_flags & 2));
  // (the kernel finds a position's phase through fp32: positions, < len + 40 dilation, must be exact there)
  FH_CHECK_ARG(len < (1 << 24) - 4096, "fh_conv_wino54_f32: rows of %d samples: at most %d", len, (1 << 24) - 4097);
  if (tile_cfg & FH_WINO_BF16X6) {
    const int id = tile_cfg & ~FH_WINO_BF16X6;
    FH_CHECK_ARG(id == 1 || id == 2, "fh_conv_wino54_f32: tile_cfg %d has no bf16 x 6 form (96- and 64-row blocks only: 1, 2)", id);
    return fh_internal_wino54_bf(groups, n_groups, batch, cout_pad, len, dilation, pm, 4 - id, vl, st, run_map, n_runs);
  }
#define FH_W54_CASE(id, MT)                                                                                         \
  case id:                                                                                                          \
    return vl ? launch_wino54<MT, true>(groups, n_groups, batch, cout_pad, len, dilation, pm, st, run_map, n_runs)  \
              : launch_wino54<MT, false>(groups, n_groups, batch, cout_pad, len, dilation, pm, st, run_map, n_runs);
  switch (tile_cfg) {
    FH_W54_CASE(0, 4)
    FH_W54_CASE(1, 3)
    FH_W54_CASE(2, 2)
    case 3:
      return vl ? launch_wino54<2, true, true>(groups, n_groups, batch, cout_pad, len, dilation, pm, st, run_map, n_runs)
                : launch_wino54<2, false, true>(groups, n_groups, batch, cout_pad, len, dilation, pm, st, run_map, n_runs);
  }
#undef FH_W54_CASE
  fh_set_error("fh_conv_wino54_f32: unknown tile_cfg %d", tile_cfg);
  return FH_E_ARG;
}
}  // namespace

extern "C" int fh_conv_wino54_f32(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len,
                                  int dilation, int phase_major, int tile_cfg, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && batch > 0 && len > 0, "fh_conv_wino54_f32: bad sizes");
  FH_CHECK_ARG(dilation >= 1 && dilation <= 64, "fh_conv_wino54_f32: dilation %d unsupported", dilation);
  return wino54_dispatch(groups, n_groups, batch, cout_pad, len, dilation, phase_major != 0 ? 1 : 0, tile_cfg, (hipStream_t)stream,
                         nullptr, 0);
}

extern "C" int fh_wino54_n_tiles(int len, int dilation, int phase_major) {
  return (len > 0 && dilation >= 1) ? wino54_n_tiles(len, dilation, phase_major != 0) : -1;
}

extern "C" int fh_wino54_run_len(int n_tiles) { return n_tiles > 0 ? fh_cdiv(n_tiles, fh_cdiv(n_tiles, V_RUN)) : -1; }

extern "C" int fh_conv_wino54_ragged_f32(const fh_wino_group* groups, int n_groups, int cout_pad, int max_len, int dilation,
                                         int layout_flags, int tile_cfg, const int* run_map, int n_runs, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && max_len > 0 && run_map && n_runs > 0, "fh_conv_wino54_ragged_f32: bad sizes");
  FH_CHECK_ARG(layout_flags >= 0 && layout_flags <= 3, "fh_conv_wino54_ragged_f32: layout_flags %d (bit 0 phase-major, bit 1 no vector accesses)", layout_flags);
  FH_CHECK_ARG(dilation >= 1 && dilation <= 64, "fh_conv_wino54_ragged_f32: dilation %d unsupported", dilation);
  return wino54_dispatch(groups, n_groups, 1, cout_pad, max_len, dilation, layout_flags, tile_cfg, (hipStream_t)stream, run_map, n_runs);
}
