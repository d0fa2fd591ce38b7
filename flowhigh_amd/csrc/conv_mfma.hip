// Grouped implicit-GEMM 1-D convolution on the fp32 matrix cores of gfx950.
//
// Replaces the Conv1d / ConvTranspose1d call sites of BigVGAN
// (/root/reference/src/flowhigh/models/bigvgan/models.py:63-72,172-194).
//
// GEMM view per group:  D[co, n] = sum_{seg} sum_{tap} sum_{ci} W[tap, ci, co] * X[ci, n + off(tap)]
//   M = output channels, N = time positions, K = (segment, channel chunk, tap, channel).
// v_mfma_f32_32x32x2_f32: A lane l = A[i = l & 31][k = l >> 5], B lane l = B[k = l >> 5][j = l & 31],
// D reg r of lane l = D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]  (exact fp32, fma chain).
//
// Block = 4 waves, tile BM x BN = (32 MT WM) x (32 NT WN).  K is walked in steps of
// (CK input channels) x (one tap), CK = 16 (or 8 when cin % 16 != 0):
//   * the CK-channel input slab [CK][BN + halo] is staged in LDS once per channel chunk and every
//     tap reads it at a shifted column (the B fragment of a tap is a shifted ds_read_b32, lanes on
//     consecutive addresses: conflict free).  Rows are fetched with raw buffer loads whose
//     descriptor spans exactly one (batch, channel) row, so the conv's zero padding is the
//     hardware's out-of-range-returns-0 (no per-element branches);
//   * the weight tile [BM][CK] of a (chunk, tap) is a contiguous block of the packed weights; rows
//     are padded to CK + 4 floats in LDS so that one ds_read_b128 per lane (conflict free) yields
//     the A values of four k-steps.  K order inside a chunk is permuted (k-step (q, e) pairs
//     channels 8q + e and 8q + 4 + e); A and B use the same permutation;
//   * both are double buffered: loads of step i+1 are issued before the MFMAs of step i and
//     written to the other LDS buffer after them; one barrier per step (CK/2 * MT * NT MFMAs per
//     wave).  The tap offset of step i+1 is fetched (scalar) during step i.
// Block -> work mapping is XCD aware: blocks with equal (id mod 8) share an L2; the n-tiles of one
// (group, batch, co-tile) panel are dealt to one XCD in runs of 8, so the weight tiles they share
// are fetched into that L2 once.
#include "fh_common.h"

namespace {

constexpr int NT_RUN = 8;       // n-tiles of a panel that run together on one XCD

template <int MT, int NT, int WM, int WN, int CK>
struct ConvCfg {
  static constexpr int BM = 32 * MT * WM;
  static constexpr int BN = 32 * NT * WN;
  static constexpr int WP = CK + 4;                      // LDS pitch of a weight row, floats
  static constexpr int XW = BN + FH_CONV_MAX_HALO;       // staged columns per channel
  static constexpr int XROWS = CK / 4;                   // rows staged by each wave
  static constexpr int XREG = (XW + 63) / 64;            // dwords per lane per row
  static constexpr int WF4 = BM * CK / 4;                // float4s in a weight tile
  static constexpr int WREG = (WF4 + 255) / 256;
  static constexpr int LDS_FLOATS = 2 * BM * WP + 2 * CK * XW;
};

struct SegU {          // wave-uniform copy of the hot fields of one fh_conv_seg
  const float* x;
  const float* w;
  int cin, ntaps, off_min;
};
__device__ __forceinline__ SegU load_seg(const fh_conv_seg* S) {
  SegU u;
  u.x = uni(S->x);
  u.w = uni(S->w);
  u.cin = uni(S->cin);
  u.ntaps = uni(S->ntaps);
  u.off_min = uni(S->off_min);
  return u;
}

// PH > 1: "phase-fused" groups (fh_conv_transpose_fused_f32): the group's PH segments are the PH output phases of a
// ConvTranspose1d with stride PH (the same input, each phase its own taps and weights): segment s accumulates into its OWN
// accumulator set and the epilogue stores the PH phases of an input position as PH consecutive floats.  One block then writes
// whole lines; with one block per phase every 128-byte line of the output was written by PH blocks in PH pieces (strided
// 4-byte stores: 130 MB written for 46 MB of output over a step's upsamplers).  The arithmetic of a phase is unchanged.
template <int MT, int NT, int WM, int WN, int CK, int PH = 1>
// (blocks per CU the launch bounds promise: 3 for the small accumulator sets, except the 32 x 256 tile with 16-channel chunks, whose
// staging registers leave room for 2 -- the compiler said so with -Wpass-failed until round 6)
__global__ __launch_bounds__(256, (MT * NT * PH <= 4 && !(MT == 1 && NT == 4 && CK == 16) ? 3 : (MT * NT * PH <= 8 ? 2 : 1))) void conv_mfma_kernel(
    const fh_conv_group* __restrict__ groups, int n_groups, int batch, int co_tiles, int n_tiles) {
  using Cfg = ConvCfg<MT, NT, WM, WN, CK>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, XW = Cfg::XW, WP = Cfg::WP;
  constexpr int XROWS = Cfg::XROWS, XREG = Cfg::XREG, WREG = Cfg::WREG, KQ = CK / 8;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS + FH_CONV_MAX_SEG * FH_CONV_MAX_TAPS];
  float* ws = lds;
  float* xs = lds + 2 * BM * WP;
  int* toff = reinterpret_cast<int*>(lds + Cfg::LDS_FLOATS);   // [seg][tap] column shift of each tap

  // ---- block -> (panel, n tile); panels = (group, batch, co tile), heavy groups first ----
  const int panels = n_groups * batch * co_tiles;
  const int runs_per_panel = (n_tiles + NT_RUN - 1) / NT_RUN;
  const int total_runs = panels * runs_per_panel;
  const int bid = blockIdx.x;
  const int slot = bid >> 3;
  const int run = (slot / NT_RUN) * 8 + (bid & 7);
  if (run >= total_runs) return;
  // (runtime integer divisions are done on the VALU: pin the results back to SGPRs)
  const int panel = uni(run / runs_per_panel);
  const int ntile = uni((run % runs_per_panel) * NT_RUN + (slot % NT_RUN));
  if (ntile >= n_tiles) return;
  const int cot = uni(panel % co_tiles);
  const int gb = uni(panel / co_tiles);
  const int b = uni(gb % batch);
  const fh_conv_group* __restrict__ G = groups + uni(gb / batch);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int co0 = cot * BM;
  const int n0 = ntile * BN;
  const int lin = uni(G->lin), cout_pad = uni(G->cout_pad), nseg = uni(G->nseg);
  // groups of one launch may have different lengths (ragged batches: one group per clip, the grid is sized for the
  // longest): a block past its group's last column has nothing to do
  if (n0 >= uni(G->n_len)) return;

  f32x16 acc[PH][MT][NT];
#pragma unroll
  for (int p = 0; p < PH; ++p)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][i][j][r] = 0.f;

  // total number of (chunk, tap) steps over all segments
  int nsteps = 0;
  for (int s = 0; s < nseg; ++s) nsteps += uni((G->seg[s].cin / CK) * G->seg[s].ntaps);

  // tap shifts relative to the staged slab, read back (LDS broadcast) one step ahead of their use
  if (tid < FH_CONV_MAX_SEG * FH_CONV_MAX_TAPS) {
    const fh_conv_seg* sg = &G->seg[tid / FH_CONV_MAX_TAPS];
    toff[tid] = (tid / FH_CONV_MAX_TAPS) < nseg ? sg->tap_off[tid % FH_CONV_MAX_TAPS] - sg->off_min : 0;
  }

  // Two register sets for weight tiles: the tile of step i+2 is requested while step i computes
  // and is written to LDS at the end of step i+1, so a load has two full MFMA phases to land (a wave
  // that is alone on its SIMD at the tail of a launch would otherwise stall on every tile).
  u32x4 wregA[WREG], wregB[WREG];
  unsigned xreg[XROWS][XREG];

  // ---- cursors ------------------------------------------------------------------------------------
  // Steps run over (segment, chunk, tap) in that order; three cursors walk them, each with a few scalar adds per step
  // (rebuilding every address from (segment, chunk, tap) cost 50-100 scalar instructions per 16-MFMA step):
  //   K  the step that computes: its tap (LDS shift), whether it is a chunk's first / last tap;
  //   W  the step whose weight tile is requested next (two ahead): the tiles of a segment are contiguous in step order,
  //      [ci/CK][tap][cout_pad][CK], so one descriptor per segment and a scalar byte offset that grows by one tile row
  //      block (cout_pad CK floats) per step;
  //   X  the chunk whose slab is requested next (one ahead): pointer to this wave's first row of it.
  // A segment's fields are read when a cursor enters it.
  struct Cur { int s, j, nt, cl; };          // segment, tap, taps per chunk, chunks left in the segment (this one included)
  Cur K = {0, 0, uni(G->seg[0].ntaps), uni(G->seg[0].cin) / CK};
  const unsigned wstep = (unsigned)cout_pad * CK * 4u;
  int w_seg = -1, w_left = 0;
  unsigned w_soff = 0;
  __amdgpu_buffer_rsrc_t w_r = make_rsrc(nullptr, 0);
  auto w_enter = [&]() {                       // W cursor: first step of the next segment
    ++w_seg;
    const fh_conv_seg* sg = &G->seg[w_seg < nseg ? w_seg : nseg - 1];
    w_left = w_seg < nseg ? (uni(sg->cin) / CK) * uni(sg->ntaps) : 0x7fffffff;
    w_r = make_rsrc(uni(sg->w) + (size_t)co0 * CK, w_seg < nseg ? (unsigned)w_left * wstep - (unsigned)co0 * CK * 4u : 0u);
    w_soff = 0;
  };
  unsigned wvoff[WREG];                        // lanes beyond the tile request nothing (out-of-range offset, never stored)
#pragma unroll
  for (int i = 0; i < WREG; ++i) wvoff[i] = tid + 256 * i < Cfg::WF4 ? (unsigned)(tid + 256 * i) * 16u : 0x80000000u;
  auto load_w = [&](u32x4 (&wreg)[WREG]) {     // the W cursor's tile, then on to the next step
#pragma unroll
    for (int i = 0; i < WREG; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(w_r, wvoff[i], w_soff, 0);
    w_soff += wstep;
    if (--w_left == 0) w_enter();
  };
  auto store_w = [&](const u32x4 (&wreg)[WREG], int buf) {
    float* dst = ws + buf * BM * WP;
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int f = tid + 256 * i;
      if (f < Cfg::WF4) *reinterpret_cast<u32x4*>(dst + (f / (CK / 4)) * WP + 4 * (f % (CK / 4))) = wreg[i];
    }
  };
  int x_seg = -1, x_left = 0, x_t0 = 0;
  const float* x_row = nullptr;                // first row of the X cursor's chunk that this wave stages
  auto x_enter = [&]() {
    ++x_seg;
    const fh_conv_seg* sg = &G->seg[x_seg < nseg ? x_seg : nseg - 1];
    const int cin = uni(sg->cin);
    x_left = x_seg < nseg ? cin / CK : 0x7fffffff;
    x_row = uni(sg->x) + ((size_t)b * cin + wave * XROWS) * lin;
    x_t0 = n0 + uni(sg->off_min);
  };
  auto load_x = [&]() {                        // the X cursor's chunk, then on to the next one
#pragma unroll
    for (int rr = 0; rr < XROWS; ++rr) {
      __amdgpu_buffer_rsrc_t r = make_rsrc(x_row + (size_t)rr * lin, (unsigned)lin * 4u);
#pragma unroll
      for (int i = 0; i < XREG; ++i)   // t < 0 wraps to a huge unsigned offset: out of range -> 0
        xreg[rr][i] = __builtin_amdgcn_raw_buffer_load_b32(r, (x_t0 + lane + 64 * i) * 4, 0, 0);
    }
    x_row += (size_t)CK * lin;
    if (--x_left == 0) x_enter();
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int rr = 0; rr < XROWS; ++rr) {
      float* dst = xs + buf * CK * XW + (wave * XROWS + rr) * XW;
#pragma unroll
      for (int i = 0; i < XREG; ++i) {
        const int j = lane + 64 * i;
        if (j < XW) dst[j] = __uint_as_float(xreg[rr][i]);
      }
    }
  };

  // ---- prologue ----------------------------------------------------------------------------
  int wbuf = 0, xbuf = 0;
  w_enter();
  x_enter();
  load_w(wregA);
  load_x();
  if (nsteps > 1) load_w(wregB);                       // tile of step 1, stored at the end of step 0
  store_w(wregA, 0);
  store_x(0);
  __syncthreads();

  // one K step.  LOAD: register set that receives the tile of step it+2; STORE: the set holding
  // the tile of step it+1 (requested one step earlier).  Everything that is not an MFMA is issued
  // in the shadow of the MFMAs of this step (pinned with sched_barrier): global prefetch after
  // k-step 1, LDS stores of the next tiles at mid-step, so that a wave that is alone on its SIMD
  // keeps the matrix pipe fed; only the barrier and the first fragment reads stay exposed.
  int xoff_cur = toff[0];
  auto step = [&](f32x16 (&ACC)[MT][NT], int it, u32x4 (&LOAD)[WREG], const u32x4 (&STORE)[WREG]) {
    constexpr int KS = 4 * KQ;
    const bool last_tap = K.j == K.nt - 1;
    const bool more_chunks = K.cl > 1 || K.s + 1 < nseg;
    const bool flip_x = last_tap && more_chunks;
    // (segment, tap) of the next step, for its LDS shift
    const int j1 = last_tap ? 0 : K.j + 1;
    const int s1 = (last_tap && K.cl == 1 && K.s + 1 < nseg) ? K.s + 1 : K.s;
    const float* wsb = ws + wbuf * BM * WP;
    const float* xsb = xs + xbuf * CK * XW + xoff_cur + wn * NT * 32 + l31 + 4 * lh * XW;
    f32x4 a[KQ][MT];
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        a[q][mt] = *reinterpret_cast<const f32x4*>(wsb + ((wm * MT + mt) * 32 + l31) * WP + 4 * (2 * q + lh));
    float bf[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[0][nt] = xsb[nt * 32];
    int xoff_next = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {            // k-step ks = 4 q + e uses slab row 8 q + 4 lh + e
      if (ks + 1 < KS) {
        const int q1 = (ks + 1) >> 2, e1 = (ks + 1) & 3;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[(ks + 1) & 1][nt] = xsb[(8 * q1 + e1) * XW + nt * 32];
      }
      if (ks == 1) {                              // global prefetch
        if (it + 2 < nsteps) load_w(LOAD);
        if (K.j == 0 && more_chunks) load_x();    // slab of the next chunk, stored after this chunk's last tap
      }
      if (ks == KS / 2) {                         // LDS stores of the tiles of step it+1
        if (it + 1 < nsteps) store_w(STORE, wbuf ^ 1);
        if (flip_x) store_x(xbuf ^ 1);
      }
      if (ks == KS - 2) xoff_next = toff[s1 * FH_CONV_MAX_TAPS + j1];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          ACC[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks >> 2][mt][ks & 3], bf[ks & 1][nt],
                                                             ACC[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    wbuf ^= 1;
    if (flip_x) xbuf ^= 1;
    xoff_cur = xoff_next;
    if (last_tap) {
      K.j = 0;
      if (--K.cl == 0 && K.s + 1 < nseg) {
        ++K.s;
        K.nt = uni(G->seg[K.s].ntaps);
        K.cl = uni(G->seg[K.s].cin) / CK;
      }
    } else {
      ++K.j;
    }
  };

  if constexpr (PH == 1) {
    for (int it = 0; it < nsteps; it += 2) {
      step(acc[0], it, wregA, wregB);
      if (it + 1 < nsteps) step(acc[0], it + 1, wregB, wregA);
    }
  } else {
    // one loop per phase (its accumulator set is a compile-time choice); every phase has an even number of steps (checked by
    // the host), so the two weight register sets keep their roles across the phase boundaries
    int it = 0;
#pragma unroll
    for (int p = 0; p < PH; ++p) {
      const int n = uni((G->seg[p].cin / CK) * G->seg[p].ntaps);
      for (int j = 0; j < n; j += 2, it += 2) {
        step(acc[p], it, wregA, wregB);
        step(acc[p], it + 1, wregB, wregA);
      }
    }
  }

  // ---- epilogue: bias + residuals, scale, strided store -------------------------------------
  // One buffer descriptor per tensor spans this batch item's [cout, lout] slab, so rows past
  // cout fall out of range by themselves; columns past n_len get an out-of-range offset.  Every
  // load and store is then unconditional (no exec-mask branches) and can be issued in bulk.
  const int nres = uni(G->nres);
  const float scale = G->scale;
  const int cout = uni(G->cout), lout = uni(G->lout), n_len = uni(G->n_len);
  const int ostride = uni(G->out_stride), ophase = uni(G->out_phase);
  const float* __restrict__ bias = uni(G->bias);
  const size_t slab = (size_t)b * cout * lout;
  const unsigned slab_bytes = (unsigned)cout * (unsigned)lout * 4u;
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(uni((const float*)G->out) + slab, slab_bytes);
  const __amdgpu_buffer_rsrc_t rr0 = make_rsrc(nres > 0 ? uni(G->res[0]) + slab : nullptr, nres > 0 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rr1 = make_rsrc(nres > 1 ? uni(G->res[1]) + slab : nullptr, nres > 1 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rr2 = make_rsrc(nres > 2 ? uni(G->res[2]) + slab : nullptr, nres > 2 ? slab_bytes : 0u);
  unsigned coloff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + (wn * NT + nt) * 32 + l31;
    coloff[nt] = n < n_len ? (unsigned)(n * ostride + ophase) * 4u : 0x80000000u;
  }
  if constexpr (PH > 1) {
    // out[co, PH n + p] for p < PH: PH consecutive floats per lane, lanes on consecutive positions: whole lines per wave store
    // (ostride == PH, ophase == 0, no residuals: checked by the launcher / the host)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float bv = (bias && co < cout) ? bias[co] : 0.f;
        const unsigned rowoff = (unsigned)co * (unsigned)lout * 4u;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const unsigned off = (co < cout) ? rowoff + coloff[nt] : 0x80000000u;
          if constexpr (PH == 2) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = {__float_as_uint((acc[0][mt][nt][r] + bv) * scale), __float_as_uint((acc[1][mt][nt][r] + bv) * scale)};
            __builtin_amdgcn_raw_buffer_store_b64(v, ro, off, 0, 0);
          } else {
            typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
            const u32x3 v = {__float_as_uint((acc[0][mt][nt][r] + bv) * scale), __float_as_uint((acc[1][mt][nt][r] + bv) * scale),
                             __float_as_uint((acc[PH - 1][mt][nt][r] + bv) * scale)};
            __builtin_amdgcn_raw_buffer_store_b96(v, ro, off, 0, 0);
          }
        }
      }
    return;
  }
  auto& acc1 = acc[0];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float v[4][NT];
      unsigned off[4][NT];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co0 + (wm * MT + mt) * 32 + q + 8 * g4 + 4 * lh;
        const float bv = (bias && co < cout) ? bias[co] : 0.f;
        const unsigned rowoff = (unsigned)co * (unsigned)lout * 4u;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          off[q][nt] = (co < cout) ? rowoff + coloff[nt] : 0x80000000u;
          v[q][nt] = acc1[mt][nt][4 * g4 + q] + bv;
        }
      }
      if (nres > 0) {
        float t0[4][NT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            t0[q][nt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr0, off[q][nt], 0, 0));
        if (nres > 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              t0[q][nt] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr1, off[q][nt], 0, 0));
        }
        if (nres > 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              t0[q][nt] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr2, off[q][nt], 0, 0));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) v[q][nt] += t0[q][nt];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q][nt] * scale), ro, off[q][nt], 0, 0);
    }
  }
}

template <int MT, int NT, int WM, int WN, int CK, int PH = 1>
int launch_conv(const fh_conv_group* groups, int n_groups, int batch, int cout_pad, int n_len,
                hipStream_t stream) {
  using Cfg = ConvCfg<MT, NT, WM, WN, CK>;
  const int co_tiles = cout_pad / Cfg::BM;
  const int n_tiles = fh_cdiv(n_len, Cfg::BN);
  const long long panels = (long long)n_groups * batch * co_tiles;
  const long long runs = panels * fh_cdiv(n_tiles, NT_RUN);
  const long long blocks = (long long)fh_cdiv(runs, 8) * 8 * NT_RUN;
  FH_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "fh_conv_grouped_f32: grid too large");
  hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, WM, WN, CK, PH>), dim3((unsigned)blocks), dim3(256), 0, stream,
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

// len % 4 == 0 (rows 16-byte aligned): 4 consecutive outputs per thread from 3 aligned float4 loads per channel
// (7 taps: x[t-3 .. t+6] lies inside x[t-4 .. t+7]) instead of 7 dword loads per output.
__global__ __launch_bounds__(256) void conv_post_tanh_vec_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ out, int cin, int len) {
  extern __shared__ float wl[];   // cin * 7
  for (int i = threadIdx.x; i < cin * 7; i += 256) wl[i] = w[i];
  __syncthreads();
  const int b = blockIdx.y;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= len) return;
  const float* xb = x + (size_t)b * cin * len;
  const float b0 = bias[0];
  f32x4 acc = {b0, b0, b0, b0};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < cin; ++c) {
    const float* xr = xb + (size_t)c * len + t;
    const f32x4 lo = t >= 4 ? *reinterpret_cast<const f32x4*>(xr - 4) : zero;
    const f32x4 mid = *reinterpret_cast<const f32x4*>(xr);
    const f32x4 hi = t + 4 < len ? *reinterpret_cast<const f32x4*>(xr + 4) : zero;
    const float v[12] = {lo[0], lo[1], lo[2], lo[3], mid[0], mid[1], mid[2], mid[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) {
      const float wv = wl[c * 7 + jj];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaf(wv, v[r + jj + 1], acc[r]);     // x[t + r + jj - 3] = v[r + jj + 1]
    }
  }
  const f32x4 o = {tanhf(acc[0]), tanhf(acc[1]), tanhf(acc[2]), tanhf(acc[3])};
  *reinterpret_cast<f32x4*>(out + (size_t)b * len + t) = o;
}

struct TileInfo { int bm, bn; };
constexpr TileInfo kTiles[] = {{128, 128}, {192, 128}, {96, 256}, {64, 256}, {32, 512}, {128, 64}, {96, 128}};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

}  // namespace

extern "C" int fh_sizeof_conv_group(void) { return (int)sizeof(fh_conv_group); }

extern "C" int fh_conv_tile_m(int cfg) { return (cfg >= 0 && cfg < kNumTiles) ? kTiles[cfg].bm : -1; }
extern "C" int fh_conv_tile_n(int cfg) { return (cfg >= 0 && cfg < kNumTiles) ? kTiles[cfg].bn : -1; }

extern "C" int fh_conv_grouped_f32(const fh_conv_group* groups, int n_groups, int batch,
                                   int cout_pad, int n_len, int tile_cfg, int ck, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && batch > 0 && n_len > 0, "fh_conv_grouped_f32: bad sizes");
  const int bm = fh_conv_tile_m(tile_cfg);
  FH_CHECK_ARG(bm > 0, "fh_conv_grouped_f32: unknown tile_cfg %d", tile_cfg);
  FH_CHECK_ARG(cout_pad % bm == 0, "fh_conv_grouped_f32: cout_pad %d not a multiple of tile %d", cout_pad, bm);
  FH_CHECK_ARG(ck == 8 || ck == 16, "fh_conv_grouped_f32: channel chunk must be 8 or 16 (got %d)", ck);
  // per-clip tensors are addressed with 32-bit byte offsets (buffer descriptors): cout * lout * 4 < 2^31
  // is checked by the host plan (flowhigh_amd/vocoder.py) where the shapes are known.
  hipStream_t st = (hipStream_t)stream;
#define FH_CONV_CASE(id, MT, NT, WM, WN)                                                            \
  case id:                                                                                          \
    return ck == 16 ? launch_conv<MT, NT, WM, WN, 16>(groups, n_groups, batch, cout_pad, n_len, st) \
                    : launch_conv<MT, NT, WM, WN, 8>(groups, n_groups, batch, cout_pad, n_len, st);
  switch (tile_cfg) {
    FH_CONV_CASE(0, 2, 2, 2, 2)
    FH_CONV_CASE(1, 3, 2, 2, 2)
    FH_CONV_CASE(2, 3, 2, 1, 4)
    FH_CONV_CASE(3, 2, 2, 1, 4)
    FH_CONV_CASE(4, 1, 4, 1, 4)
    FH_CONV_CASE(5, 2, 1, 2, 2)
    FH_CONV_CASE(6, 3, 1, 1, 4)
  }
#undef FH_CONV_CASE
  return FH_E_ARG;
}

extern "C" int fh_conv_transpose_fused_f32(const fh_conv_group* groups, int n_groups, int batch, int cout_pad, int n_len,
                                          int tile_cfg, int phases, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && batch > 0 && n_len > 0, "fh_conv_transpose_fused_f32: bad sizes");
  FH_CHECK_ARG(phases == 2, "fh_conv_transpose_fused_f32: %d phases (2: the stride-3 form left the library with ABI 4)", phases);
  const int bm = fh_conv_tile_m(tile_cfg);
  FH_CHECK_ARG(bm > 0 && cout_pad % bm == 0, "fh_conv_transpose_fused_f32: cout_pad %d / tile_cfg %d", cout_pad, tile_cfg);
  hipStream_t st = (hipStream_t)stream;
#define FH_CONVT_CASE(id, MT, NT, WM, WN) \
  case id:                                \
    return launch_conv<MT, NT, WM, WN, 16, 2>(groups, n_groups, batch, cout_pad, n_len, st);
  switch (tile_cfg) {          // the shapes the launch plans use for upsamplers (channel chunk 16)
    FH_CONVT_CASE(3, 2, 2, 1, 4)
    FH_CONVT_CASE(4, 1, 4, 1, 4)
    FH_CONVT_CASE(6, 3, 1, 1, 4)
  }
#undef FH_CONVT_CASE
  fh_set_error("fh_conv_transpose_fused_f32: tile_cfg %d has no phase-fused form (3, 4, 6)", tile_cfg);
  return FH_E_ARG;
}

extern "C" int fh_conv_post_tanh_f32(const float* x, const float* w, const float* bias, float* out,
                                     int batch, int cin, int len, int ksz, void* stream) {
  FH_CHECK_ARG(x && w && bias && out && batch > 0 && cin > 0 && len > 0, "fh_conv_post_tanh_f32: bad args");
  FH_CHECK_ARG((ksz & 1) && ksz <= 15, "fh_conv_post_tanh_f32: ksz %d unsupported", ksz);
  if (ksz == 7 && len % 4 == 0 && ((((size_t)x) | ((size_t)out)) & 15) == 0) {
    hipLaunchKernelGGL(conv_post_tanh_vec_kernel, dim3(fh_cdiv(len, 1024), batch), dim3(256), cin * 7 * sizeof(float),
                       (hipStream_t)stream, x, w, bias, out, cin, len);
  } else {
    dim3 grid(fh_cdiv(len, 256), batch);
    hipLaunchKernelGGL(conv_post_tanh_kernel, grid, dim3(256), cin * ksz * sizeof(float),
                       (hipStream_t)stream, x, w, bias, out, cin, len, ksz);
  }
  FH_CHECK_LAUNCH("fh_conv_post_tanh_f32");
  return FH_OK;
}
