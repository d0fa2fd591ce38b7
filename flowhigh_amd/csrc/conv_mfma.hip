// Grouped implicit-GEMM 1-D convolution on the fp32 matrix cores of gfx950.
//
// Replaces the Conv1d / ConvTranspose1d call sites of BigVGAN
// (/root/reference/src/flowhigh/models/bigvgan/models.py:63-72,172-194).
//
// GEMM view per group:  D[co, n] = sum_{seg} sum_{tap} sum_{ci} W[tap, ci, co] * X[ci, n + off(tap)]
//   M = output channels, N = time positions, K = (segment, tap, input channel).
// v_mfma_f32_32x32x2_f32: A lane l = A[i = l & 31][k = l >> 5], B lane l = B[k = l >> 5][j = l & 31],
// D reg r of lane l = D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]  (exact fp32, fma chain).
//
// Block = 4 waves, tile BM x BN = (32 MT WM) x (32 NT WN).  K is walked in steps of
// (8 input channels) x (one tap):
//   * the 8-channel input slab [8][BN + halo] is staged in LDS once per channel chunk and every
//     tap reads it at a shifted column (the B fragment of a tap is a shifted ds_read_b32, lanes
//     on consecutive addresses: conflict free);
//   * the weight tile [BM][8] of that (chunk, tap) is a contiguous 32*BM-byte block of the packed
//     weights; rows are padded to 12 floats in LDS so that one ds_read_b128 per lane (conflict
//     free: start bank 4 (3 i mod 16)) yields the A values of four k-steps.  K order inside a chunk
//     is permuted (k-step e pairs channels e and 4 + e); A and B use the same permutation.
//   * both are double buffered: global loads of step i+1 are issued before the MFMAs of step i
//     and written to the other LDS buffer after them; one barrier per step.
// Block -> work mapping is XCD aware: blocks with equal (id mod 8) share an L2; the n-tiles of one
// (group, batch, co-tile) panel are dealt to one XCD in runs of 8, so the weight tiles they share
// are fetched into that L2 once.
#include "fh_common.h"

namespace {

constexpr int WP = 12;          // LDS pitch of a weight row (8 channels + 4 pad), floats
constexpr int NT_RUN = 8;       // n-tiles of a panel that run together on one XCD

template <int MT, int NT, int WM, int WN>
struct ConvCfg {
  static constexpr int BM = 32 * MT * WM;
  static constexpr int BN = 32 * NT * WN;
  static constexpr int XW = BN + FH_CONV_MAX_HALO;     // staged columns per channel
  static constexpr int XREG = (XW + 31) / 32;          // staging registers per thread (x slab)
  static constexpr int WREG = (BM * 2 + 255) / 256;    // staging float4s per thread (weight tile)
  static constexpr int LDS_FLOATS = 2 * BM * WP + 2 * 8 * XW;
};

template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const fh_conv_group* __restrict__ groups,
                                                        int n_groups, int batch, int co_tiles,
                                                        int n_tiles) {
  using Cfg = ConvCfg<MT, NT, WM, WN>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, XW = Cfg::XW, XREG = Cfg::XREG, WREG = Cfg::WREG;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* ws = lds;
  float* xs = lds + 2 * BM * WP;

  // ---- block -> (panel, n tile); panels = (group, batch, co tile), heavy groups first ----
  const int panels = n_groups * batch * co_tiles;
  const int runs_per_panel = (n_tiles + NT_RUN - 1) / NT_RUN;
  const int total_runs = panels * runs_per_panel;
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int slot = bid >> 3;
  const int run = (slot / NT_RUN) * 8 + xcd;
  if (run >= total_runs) return;
  const int panel = run / runs_per_panel;
  const int ntile = (run % runs_per_panel) * NT_RUN + (slot % NT_RUN);
  if (ntile >= n_tiles) return;
  const int cot = panel % co_tiles;
  const int gb = panel / co_tiles;
  const int b = gb % batch;
  const fh_conv_group& G = groups[gb / batch];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int co0 = cot * BM;
  const int n0 = ntile * BN;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging registers
  f32x4 wreg[WREG];
  float xreg[XREG];
  const int xrow = tid >> 5;      // channel row (0..7) this thread stages
  const int xcol = tid & 31;

  // ---- loaders -------------------------------------------------------------------------
  auto load_w = [&](const fh_conv_seg& S, int chunk, int tap) {
    const float* wp = S.w + ((size_t)(chunk * S.ntaps + tap) * G.cout_pad + co0) * 8;
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      int f = tid + 256 * i;
      if (f < BM * 2) wreg[i] = *reinterpret_cast<const f32x4*>(wp + 4 * f);
    }
  };
  auto store_w = [&](int buf) {
    float* dst = ws + buf * BM * WP;
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      int f = tid + 256 * i;
      if (f < BM * 2) *reinterpret_cast<f32x4*>(dst + (f >> 1) * WP + 4 * (f & 1)) = wreg[i];
    }
  };
  auto load_x = [&](const fh_conv_seg& S, int chunk) {
    const float* xp = S.x + ((size_t)b * S.cin + chunk * 8 + xrow) * G.lin;
    const int t0 = n0 + S.off_min;
    const int width = BN + S.off_max - S.off_min;
#pragma unroll
    for (int i = 0; i < XREG; ++i) {
      int j = xcol + 32 * i;
      int t = t0 + j;
      float v = 0.f;
      if (j < width && t >= 0 && t < G.lin) v = xp[t];
      xreg[i] = v;
    }
  };
  auto store_x = [&](int buf) {
    float* dst = xs + buf * 8 * XW + xrow * XW;
#pragma unroll
    for (int i = 0; i < XREG; ++i) {
      int j = xcol + 32 * i;
      if (j < XW) dst[j] = xreg[i];
    }
  };

  // ---- prologue: first (seg 0, chunk 0, tap 0) ---------------------------------------------
  int s = 0, c = 0, j = 0;
  int wbuf = 0, xbuf = 0;
  load_w(G.seg[0], 0, 0);
  load_x(G.seg[0], 0);
  store_w(0);
  store_x(0);
  __syncthreads();

  while (true) {
    const fh_conv_seg& S = G.seg[s];
    // next step indices
    int s2 = s, c2 = c, j2 = j + 1;
    if (j2 == S.ntaps) {
      j2 = 0;
      c2 = c + 1;
      if (c2 * 8 == S.cin) {
        c2 = 0;
        s2 = s + 1;
      }
    }
    const bool has_next = s2 < G.nseg;
    if (has_next) {
      load_w(G.seg[s2], c2, j2);
      if (j2 == 0) load_x(G.seg[s2], c2);
    }

    // ---- MFMAs of the current step ----
    {
      const float* wsb = ws + wbuf * BM * WP;
      const float* xsb = xs + xbuf * 8 * XW;
      f32x4 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        a[mt] = *reinterpret_cast<const f32x4*>(wsb + ((wm * MT + mt) * 32 + l31) * WP + 4 * lh);
      const int xoff = S.tap_off[j] - S.off_min + wn * NT * 32 + l31;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float bf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = xsb[(4 * lh + e) * XW + xoff + nt * 32];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], bf[nt], acc[mt][nt], 0, 0, 0);
      }
    }

    if (!has_next) break;
    store_w(wbuf ^ 1);
    if (j2 == 0) store_x(xbuf ^ 1);
    __syncthreads();
    wbuf ^= 1;
    if (j2 == 0) xbuf ^= 1;
    s = s2;
    c = c2;
    j = j2;
  }

  // ---- epilogue: bias + residuals, scale, strided store -------------------------------------
  const int nres = G.nres;
  const float scale = G.scale;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (co >= G.cout) continue;
      const float bv = G.bias ? G.bias[co] : 0.f;
      const size_t rowbase = ((size_t)b * G.cout + co) * G.lout;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 32 + l31;
        if (n >= G.n_len) continue;
        const size_t idx = rowbase + (size_t)n * G.out_stride + G.out_phase;
        float v = acc[mt][nt][r] + bv;
        for (int q = 0; q < nres; ++q) v += G.res[q][idx];
        G.out[idx] = v * scale;
      }
    }
  }
}

template <int MT, int NT, int WM, int WN>
int launch_conv(const fh_conv_group* groups, int n_groups, int batch, int cout_pad, int n_len,
                hipStream_t stream) {
  using Cfg = ConvCfg<MT, NT, WM, WN>;
  const int co_tiles = cout_pad / Cfg::BM;
  const int n_tiles = fh_cdiv(n_len, Cfg::BN);
  const long long panels = (long long)n_groups * batch * co_tiles;
  const long long runs = panels * fh_cdiv(n_tiles, NT_RUN);
  const long long blocks = (long long)fh_cdiv(runs, 8) * 8 * NT_RUN;
  FH_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "fh_conv_grouped_f32: grid too large");
  hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, WM, WN>), dim3((unsigned)blocks), dim3(256), 0, stream,
                     groups, n_groups, batch, co_tiles, n_tiles);
  FH_CHECK_LAUNCH("fh_conv_grouped_f32");
  return FH_OK;
}

// ---- conv_post + tanh -------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_post_tanh_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ out, int cin,
                                                             int len, int ksz) {
  extern __shared__ float wl[];   // cin * ksz
  for (int i = threadIdx.x; i < cin * ksz; i += 256) wl[i] = w[i];
  __syncthreads();
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= len) return;
  const float* xb = x + (size_t)b * cin * len;
  const int half = ksz / 2;
  float acc = bias[0];
  for (int c = 0; c < cin; ++c) {
    const float* xr = xb + (size_t)c * len;
    for (int jj = 0; jj < ksz; ++jj) {
      int tt = t + jj - half;
      float v = (tt >= 0 && tt < len) ? xr[tt] : 0.f;
      acc = fmaf(wl[c * ksz + jj], v, acc);
    }
  }
  out[(size_t)b * len + t] = tanhf(acc);
}

}  // namespace

extern "C" int fh_sizeof_conv_group(void) { return (int)sizeof(fh_conv_group); }

extern "C" int fh_conv_tile_m(int cfg) {
  switch (cfg) {
    case 0: return 128;
    case 1: return 192;
    case 2: return 96;
    case 3: return 64;
    case 4: return 32;
  }
  return -1;
}
extern "C" int fh_conv_tile_n(int cfg) {
  switch (cfg) {
    case 0: return 128;
    case 1: return 128;
    case 2: return 256;
    case 3: return 256;
    case 4: return 512;
  }
  return -1;
}

extern "C" int fh_conv_grouped_f32(const fh_conv_group* groups, int n_groups, int batch,
                                   int cout_pad, int n_len, int tile_cfg, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && batch > 0 && n_len > 0, "fh_conv_grouped_f32: bad sizes");
  const int bm = fh_conv_tile_m(tile_cfg);
  FH_CHECK_ARG(bm > 0, "fh_conv_grouped_f32: unknown tile_cfg %d", tile_cfg);
  FH_CHECK_ARG(cout_pad % bm == 0, "fh_conv_grouped_f32: cout_pad %d not a multiple of tile %d", cout_pad, bm);
  hipStream_t st = (hipStream_t)stream;
  switch (tile_cfg) {
    case 0: return launch_conv<2, 2, 2, 2>(groups, n_groups, batch, cout_pad, n_len, st);
    case 1: return launch_conv<3, 2, 2, 2>(groups, n_groups, batch, cout_pad, n_len, st);
    case 2: return launch_conv<3, 2, 1, 4>(groups, n_groups, batch, cout_pad, n_len, st);
    case 3: return launch_conv<2, 2, 1, 4>(groups, n_groups, batch, cout_pad, n_len, st);
    case 4: return launch_conv<1, 4, 1, 4>(groups, n_groups, batch, cout_pad, n_len, st);
  }
  return FH_E_ARG;
}

extern "C" int fh_conv_post_tanh_f32(const float* x, const float* w, const float* bias, float* out,
                                     int batch, int cin, int len, int ksz, void* stream) {
  FH_CHECK_ARG(x && w && bias && out && batch > 0 && cin > 0 && len > 0, "fh_conv_post_tanh_f32: bad args");
  FH_CHECK_ARG((ksz & 1) && ksz <= 15, "fh_conv_post_tanh_f32: ksz %d unsupported", ksz);
  dim3 grid(fh_cdiv(len, 256), batch);
  hipLaunchKernelGGL(conv_post_tanh_kernel, grid, dim3(256), cin * ksz * sizeof(float),
                     (hipStream_t)stream, x, w, bias, out, cin, len, ksz);
  FH_CHECK_LAUNCH("fh_conv_post_tanh_f32");
  return FH_OK;
}
