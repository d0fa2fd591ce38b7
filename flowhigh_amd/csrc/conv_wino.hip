// Winograd F(4,3) form of the residual-stack Conv1d sites of BigVGAN on the fp32 matrix cores.
//
// Replaces the AMPBlock convs (/root/reference/src/flowhigh/models/bigvgan/models.py:36-72:
// kernel 3 / 7 / 11, dilation 1 / 3 / 5, "same" padding, cin == cout).
//
// Minimal filtering, 1-D:  4 outputs of a 3-tap correlation from 6 inputs with 6 multiplies,
//   y = A^T [ (G g) .* (B^T d) ].  A k-tap filter is walked in ceil(k/3) groups of 3 taps; the
// products of all groups and all input channels are summed in the transform domain, so one conv is
// 6 independent GEMMs (one per transform point xi) of depth cin * ngrp:
//   M_xi[co, tile] = sum_{g, ci} U[g][xi][co][ci] * V_xi[ci][tile, g],
//   V_xi[ci][tile, g] = sum_j B^T[xi][j] x[ci][4 tile + 3 g + j - center],   y = A^T M.
// A dilated conv is `dilation` independent undilated convs on the decimated phases x[p + d u]:
// a block works on ONE phase, so dilation only shows up as an element stride in its global
// loads / stores.
//
// Block = 12 waves = 6 transform points x 2 halves of a 64 (co) x 128 (tiles = 512 outputs) tile
// (96 x 64 tiles for C = 96); wave (xi, th) owns M_xi for 64 co x 64 tiles.  One block per CU, 3 waves on every SIMD (a 6-wave
// block leaves the SIMDs 2/2/1/1 and the matrix pipes of two of them half idle).
//   * per wave this is the 2x2 arrangement of v_mfma_f32_32x32x2_f32 tiles of the direct kernel
//     (2 A + 2 B fragments per 4 MFMAs);
//   * A (transformed weights) goes global -> registers directly in fragment layout: lane
//     (row, half) needs U[co = row][ci = 8 half + ks] for the 8 k-steps of a 16-channel chunk,
//     i.e. 8 consecutive floats -- two 16-byte loads, fully coalesced.  One register set: the half
//     that this step's MFMAs have consumed is refilled for the next step right away;
//   * B: the raw 16-channel input slab is staged in LDS once per chunk (double buffered, one
//     barrier per CHUNK), de-interleaved into 4 planes (sample u -> plane u & 3, index u >> 2) so
//     that lane `tile` reading sample 4 tile + s is a stride-1, conflict-free read.  Wave xi forms
//     its B fragments from 3-4 such samples with its row of B^T.  On this chip every VALU
//     instruction in the loop costs matrix-pipe time, so channel pairs are interleaved in the slab
//     and the two tile columns of a wave sit 64 tiles apart: one ds_read2st64_b64 with immediate
//     offsets fetches one sample for 2 columns x 2 k-steps, and the transform is packed fp32 math
//     over the k-step pair: 4 LDS + 8 VALU instructions per 8 MFMAs;
//   * epilogue: the waves exchange their accumulators through LDS one 32x32 tile at a time,
//     every thread applies A^T, bias, residuals and the scale, and stores 4 outputs.
// MFMA work per output: 1.5 ceil(k/3) instead of k multiply-adds per channel pair (k = 3 / 7 / 11:
// 2.0x / 1.56x / 1.83x fewer matrix-core cycles).
#include "fh_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// ---- fp32 operands as three bf16 pieces (BF = true kernels) ---------------------------------------------------------
// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even; the two
// subtractions are exact in fp32): 3 x 8 significant bits.  A product a b is summed over the six leading piece
// pairs (hh, hm, mh, mm, hl, lh) by six v_mfma_f32_32x32x16_bf16 per 16-channel k-block into the SAME fp32
// accumulator; the dropped pairs (ml, lm, ll) are <= 2^-24 |a b|, below the rounding of the fp32 accumulation itself
// (tools/micro/bf16x6.hip: a 32 x 32 x 1024 product against float64: 4.17e-7 of sum |a b| for this form, 4.19e-7 for
// v_mfma_f32_32x32x2_f32).  6 bf16 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64: 0.375 of the matrix-pipe cycles.
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {          // v_cvt_pk_bf16_f32
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// Everything that runs beside the bf16 MFMAs is written one result per lane, and this file's BF instantiations are compiled
// with -fno-slp-vectorize (conv_wino_bf.hip): at plain -O3 the SLP vectoriser re-packs adjacent scalar subtractions / FMAs into
// v_pk_add_f32 / v_pk_fma_f32, which share the matrix cores' fp32 lanes -- each costs 11-13 cycles of a 32-cycle bf16 MFMA gap,
// a plain instruction issues in the MFMA's shadow (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
__device__ __forceinline__ float bf_lo(unsigned p) {                       // the low bf16 of a pair as a float
  return __uint_as_float(__builtin_amdgcn_perm(0u, p, 0x01000c0cu));
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  unsigned hp[4], mp[4], lp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    hp[i] = pack_bf16(a, b);
    // (low half -> float by v_perm_b32: from `hp << 16` the combiner makes a SECOND v_cvt_pk of (a, 0) and then the shift)
    const float ra = a - bf_lo(hp[i]), rb = b - __uint_as_float(hp[i] & 0xffff0000u);
    mp[i] = pack_bf16(ra, rb);
    const float sa = ra - bf_lo(mp[i]), sb = rb - __uint_as_float(mp[i] & 0xffff0000u);
    lp[i] = pack_bf16(sa, sb);
  }
  h = __builtin_bit_cast(bf16x8, (u32x4){hp[0], hp[1], hp[2], hp[3]});
  m = __builtin_bit_cast(bf16x8, (u32x4){mp[0], mp[1], mp[2], mp[3]});
  l = __builtin_bit_cast(bf16x8, (u32x4){lp[0], lp[1], lp[2], lp[3]});
}
constexpr int W_A3 = 24;             // floats (96 bytes) per output row and 16-channel chunk of the 3-piece weights

constexpr int W_CK = 16;             // input channels per chunk
constexpr int W_THREADS = 768;       // 12 waves = 6 transform points x 2 tile halves
constexpr int W_EP = 36;             // pitch (floats) of a COLUMN of the epilogue exchange tiles: 16-byte aligned, and
                                     // 36 col mod 64 takes 16 distinct multiples of 4 over 16 lanes: conflict-free b128
#ifndef W_RUN_N
#define W_RUN_N 8
#endif
constexpr int W_RUN = W_RUN_N;             // n-blocks of a panel that run together on one XCD

// Wave tile = (32 MT) co x (32 NT) tiles; block tile = (32 MT) co x (64 NT) tiles (256 NT outputs).
//   <2, 2>: 64 x 512 outputs  (C % 64 == 0)        <3, 1>: 96 x 256 outputs  (C = 96)
//   <2, 1>: 64 x 256 (short dilation phases)      <1, 1>: 32 x 256 (short clips: a launch of a few dozen blocks
//                                                         is bound by the K loop of ONE block, not by the chip)
//   <4, 1>: 128 x 256 (half the B-fragment work per MFMA, twice the weight bytes per block)
// SUBS = 16-channel chunks per LDS slab buffer; only 1 is instantiated (2 = one block barrier per 32 input
// channels measured 0 % and its dword loader was never finished for multi-tap-group segments).
template <int MT, int NT, int SUBS = 1>
struct WCfg {
  static constexpr int BM = 32 * MT;                    // output channels per block
  static constexpr int BT = 64 * NT;                    // F(4,3) tiles per block
  static constexpr int P = BT + 8;                      // plane pitch, floats (>= BT + 4)
  // pitch of a channel PAIR (4 planes x 2 channels interleaved); a multiple of 128 floats where the
  // two tile columns of a wave are fetched by one ds_read2st64_b64
  static constexpr int RP2 = NT == 2 ? 1152 : 8 * P;
  static constexpr int XPT = (4 * BT + 16 + 47) / 48;   // slab samples per thread (48 threads per row)
  static constexpr int SLAB = SUBS * (W_CK / 2) * RP2;  // floats per slab buffer
  static constexpr int EPI = 12 * 32 * W_EP;            // epilogue exchange, floats
  // two exchange buffers where a block has more than one 32 x 32 sub-tile per wave (one barrier per sub-tile instead of
  // two); the 32 x 256 shape keeps one: two of its blocks share a CU
  static constexpr int EBUFS = MT * NT > 1 ? 2 : 1;
  static constexpr int LDS_FLOATS = 2 * SLAB > EBUFS * EPI ? 2 * SLAB : EBUFS * EPI;
  static_assert(RP2 >= 8 * P, "planes overlap");
};

// Rows of B^T, canonical arithmetic shared by every tile shape of this kernel (rows 1/2 and 3/4 share the
// sub-expressions p, q; a conv gives the same bits whatever tile shape the launch plan picks):
//     p = fma(alpha, x[i], x[j]);  q = fma(beta, x[k], x[l]);  v = fma(gamma, q, p)
//   row 0:  4 x0 - 5 x2 + x4 = fma(4, x0, fma(-5, x2, x4))            (q = x0: beta = 0)
//   row 1: -4 x1 - 4 x2 + x3 + x4 = (x4 - 4 x2) + (x3 - 4 x1)
//   row 2:  4 x1 - 4 x2 - x3 + x4 = (x4 - 4 x2) - (x3 - 4 x1)
//   row 3: -2 x1 - x2 + 2 x3 + x4 = (x4 - x2) + 2 (x3 - x1)
//   row 4:  2 x1 - x2 - 2 x3 + x4 = (x4 - x2) - 2 (x3 - x1)
//   row 5:  4 x1 - 5 x3 + x5 = fma(4, x1, fma(-5, x3, x5))
// 3 packed instructions per fragment (the plain 4-term chain took 4).  kBtOff = sample indices (i, j, k, l).
__device__ const int kBtOff[6][4] = {{2, 4, 0, 0}, {2, 4, 1, 3}, {2, 4, 1, 3}, {2, 4, 1, 3}, {2, 4, 1, 3}, {3, 5, 1, 1}};
__device__ const float kBtCoef[6][3] = {{-5.f, 0.f, 4.f}, {-4.f, -4.f, 1.f}, {-4.f, -4.f, -1.f},
                                        {-1.f, -1.f, 2.f}, {-1.f, -1.f, -2.f}, {-5.f, 0.f, 4.f}};

struct WSeg {
  const float* x;
  const float* u;
  int cin, ngrp, center, xlen;
};
__device__ __forceinline__ WSeg load_wseg(const fh_wino_seg* S) {
  WSeg w;
  w.x = uni(S->x);
  w.u = uni(S->u);
  w.cin = uni(S->cin);
  w.ngrp = uni(S->ngrp);
  w.center = uni(S->center);
  w.xlen = uni(S->xlen);
  return w;
}

// VL: the slab is fetched with 16-byte loads (4 consecutive samples of 2 channels per thread) and written with
// 8-byte LDS stores; needs contiguous, 16-byte aligned rows: phase-major tensors, or dilation 1 and len % 4 == 0.
// BF: the transformed weights are stored as three bf16 pieces, [C_in/16][tap group][6][C_out_pad][3][16] (96 bytes per
// row and chunk: lane (row, half) reads its 8 channels of each piece as one 16-byte load), the B fragments are
// split in registers, and a 16-channel k-block is six v_mfma_f32_32x32x16_bf16 (see split8 above).  The lane <->
// channel assignment is the one of the fp32 form: lane half h owns channels 8 h .. 8 h + 7 of the chunk.
// launch constants the block -> work mapping divides by (fh_common.h: fh_fastdiv); rect_*: the xcd_ranges mapping
struct WDivs {
  fh_fastdiv run_len, runs_per_panel, co_tiles, batch, dil, rect_r, rect_cr, rect_gb;
};

template <int MT, int NT, int SUBS, bool VL, bool BF>
__global__ __attribute__((amdgpu_flat_work_group_size(W_THREADS, W_THREADS), amdgpu_waves_per_eu(3, 3)))
void conv_wino_kernel(const fh_wino_group* __restrict__ groups, int n_groups, int batch, int co_tiles,
                      int n_tiles, int run_len, int dil, int pm, const int* __restrict__ run_map, int n_runs,
                      int xcd_ranges, WDivs dv) {
  using Cfg = WCfg<MT, NT, SUBS>;
  constexpr int W_BM = Cfg::BM, W_BT = Cfg::BT, W_P = Cfg::P, W_RP2 = Cfg::RP2, W_XPT = Cfg::XPT, W_SLAB = Cfg::SLAB;
  constexpr int W_SUB = (W_CK / 2) * W_RP2;          // floats of one 16-channel chunk inside a slab buffer
  extern __shared__ __attribute__((aligned(16))) float lds[];      // Cfg::LDS_FLOATS

  // ---- block -> (panel, n block); panels = (group, batch, co tile), heavy groups first ----------
  const int panels = n_groups * batch * co_tiles;
  // (a panel's n blocks are cut into equal runs of <= W_RUN: with fixed runs of 8 and 10 blocks per
  // panel, every other XCD would get the 2-block remainders only)
  const int runs_per_panel = (int)dv.runs_per_panel.d;
  const int total_runs = panels * runs_per_panel;
  const int bid = blockIdx.x;
  const int slot = bid >> 3;
  int panel, ntile;
  if (xcd_ranges) {
    // FH_WINO_XCD_RANGES: XCD x (= block id mod 8) works on the x-th EIGHTH of the time axis of EVERY panel.  Inside
    // an XCD the blocks of a (group, batch item) are ordered in RECTANGLES of (all its co tiles) x (R consecutive time
    // tiles), R = 32 / co_tiles: the ~32 blocks that are resident on the XCD's CUs together are the co tiles of the same
    // R time tiles, so they read their common input through one L2 while it is there, and the R blocks of a co tile
    // stream that panel's weights in lockstep.  Price: every XCD fetches every weight panel (once per rectangle): the
    // mapping for launches whose weights are small beside their activations (tools/traffic_per_launch.py), chosen by
    // the host plan; groups still in launch order (heavy first).
    // (host: rect_r = R, rect_cr = co_tiles R, rect_gb = co_tiles ceil(tpx / R) R = slots per (group, batch item))
    const int tpx = (n_tiles + 7) >> 3;
    const int R = (int)dv.rect_r.d;
    const int gbi = fh_div(slot, dv.rect_gb), rem = fh_mod(slot, gbi, dv.rect_gb);
    const int rect = fh_div(rem, dv.rect_cr), w = fh_mod(rem, rect, dv.rect_cr);
    const int wq = fh_div(w, dv.rect_r);
    const int t = rect * R + fh_mod(w, wq, dv.rect_r);
    if (gbi >= n_groups * batch || t >= tpx) return;
    panel = gbi * co_tiles + wq;
    ntile = (bid & 7) * tpx + t;
    if (ntile >= n_tiles) return;
  } else {
    const int slot_run = fh_div(slot, dv.run_len);
    int run = slot_run * 8 + (bid & 7);
    // Ragged launches (groups of different lengths, grid sized for the longest): only the runs that hold real tiles
    // are launched, listed heavy-first in run_map -- otherwise the empty runs of the short clips, which fall on
    // the same XCDs for every panel (run r of a panel -> XCD (panel * runs_per_panel + r) % 8), leave the real work
    // on 2-4 of the 8 XCDs.
    if (run_map) {
      if (run >= n_runs) return;
      run = uni(run_map[run]);
    }
    if (run >= total_runs) return;
    panel = fh_div(run, dv.runs_per_panel);
    ntile = fh_mod(run, panel, dv.runs_per_panel) * run_len + fh_mod(slot, slot_run, dv.run_len);
    if (ntile >= n_tiles) return;
  }
  const int gb = fh_div(panel, dv.co_tiles);
  const int cot = fh_mod(panel, gb, dv.co_tiles);
  const int gi = fh_div(gb, dv.batch);
  const int b = fh_mod(gb, gi, dv.batch);
  const fh_wino_group* __restrict__ G = groups + gi;
  const int tb = fh_div(ntile, dv.dil);              // 256-output block within the phase
  const int ph = fh_mod(ntile, tb, dv.dil);          // phase of the decimated sequence

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave % 6;                    // transform point of this wave
  const int th = wave / 6;                    // tile half: tiles 32 th + {0..31} and 64 + 32 th + {0..31}
  const int l31 = lane & 31, lh = lane >> 5;
  const int co0 = cot * W_BM;
  const int len = uni(G->len), cout_pad = uni(G->cout_pad), nseg = uni(G->nseg);
  // Groups of one launch may have different lengths (ragged batches: one group per clip, the grid is sized for
  // the longest): a block whose first output lies past its group's row has nothing to do.
  if (tb * (4 * W_BT) * dil + ph >= len) return;

  // phase-major tensors (pm): row = dil phases of lp samples, x[p + dil u] at p * lp + u
  const int lq = fh_div(len, dv.dil), lr = fh_mod(len, lq, dv.dil);         // phase p of a row holds lq + (p < lr) samples
  const int lp = (lq + (lr > 0) + 3) & ~3;
  const int pitch = pm ? dil * lp : len;             // floats per (batch, channel) row, inputs and outputs

  const int bo0 = kBtOff[xi][0], bo1 = kBtOff[xi][1], bo2 = kBtOff[xi][2], bo3 = kBtOff[xi][3];
  const float bc0 = kBtCoef[xi][0], bc1 = kBtCoef[xi][1], bc2 = kBtCoef[xi][2];

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- loaders ------------------------------------------------------------------------------
  // Every load below is issued unconditionally from straight-line code ("nothing to load" is a
  // zero-sized descriptor: the hardware returns 0 without touching memory), so the number of loads
  // in flight at each use is a compile-time constant and the s_waitcnt the compiler places are
  // exact; with loads under branches it has to assume the worst path and drains the queue.
  // slab: thread (row = tid / 48, tt = tid % 48) stages decimated samples u = tt + 48 i of its row
  const int lrow = tid / 48, ltt = tid % 48;
  const int lds_st = (lrow >> 1) * W_RP2 + ((ltt & 3) * W_P + (ltt >> 2)) * 2 + (lrow & 1);   // + 24 i
  unsigned xreg[W_XPT];
  // (S.xlen > 0: the segment's rows are xlen samples long, not len -- a transposed-conv phase whose output has one
  // sample more than u * its input, plain layout only)
  auto load_x = [&](const WSeg& S, int chunk, bool valid) {
    const int xl = S.xlen > 0 ? S.xlen : len, xp = S.xlen > 0 ? S.xlen : pitch;
    const __amdgpu_buffer_rsrc_t r =
        make_rsrc(uni(S.x + (size_t)b * S.cin * xp), valid ? (unsigned)(S.cin * xp) * 4u : 0u);
    const int ub = tb * (4 * W_BT) - S.center;                           // first staged decimated index (uniform)
    const int posb = ub * dil + ph;                                      // ... and its position in the clip
    const int rowoff = (chunk * W_CK + lrow) * xp;
    const int estride = pm ? 1 : dil;                                    // element stride of consecutive u
    const int eoff0 = rowoff + (pm ? ph * lp + ub + ltt : posb + ltt * dil);
    if (posb >= 0 && posb + (48 * W_XPT - 1) * dil < xl) {                // block interior: no per-sample checks
#pragma unroll
      for (int i = 0; i < W_XPT; ++i)
        xreg[i] = __builtin_amdgcn_raw_buffer_load_b32(r, (unsigned)(eoff0 + 48 * i * estride) * 4u, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < W_XPT; ++i) {
        const int pos = posb + (ltt + 48 * i) * dil;     // outside the clip: out-of-range offset -> 0
        const unsigned off = (unsigned)pos < (unsigned)xl ? (unsigned)(eoff0 + 48 * i * estride) * 4u : 0x80000000u;
        xreg[i] = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
      }
    }
  };
  auto store_x = [&](int buf, int sub) {
    float* dst = lds + buf * W_SLAB + sub * W_SUB + lds_st;
#pragma unroll
    for (int i = 0; i < W_XPT; ++i) dst[24 * i] = __uint_as_float(xreg[i]);
  };
  // VL form: thread (pair = tid / 96, vt = tid % 96) handles absolute quads q0 + vt (+ 96): samples
  // 4 Q .. 4 Q + 3 of its two channels = two aligned 16-byte loads, then 4 unmasked ds_write_b64 (channel pair)
  // into the 4 planes.  The slab then starts at the quad boundary below the first sample needed; the B reads
  // add the 0-3 samples of slack (`sh`) to their sample index.
  constexpr int W_XQ = W_BT + 5;                       // absolute quads a slab can touch
  constexpr int W_NITEM = (W_XQ + 95) / 96;
  const int vp = tid / 96, vt = tid % 96;
  u32x4 xq[2][2];                                      // [item slot][channel of the pair]
  auto vl_load = [&](const WSeg& S, int chunk, bool valid, int item, int slot) {
    const int xl = S.xlen > 0 ? S.xlen : len, xp = S.xlen > 0 ? S.xlen : pitch;      // (xlen % 4 == 0 here: host)
    const __amdgpu_buffer_rsrc_t r =
        make_rsrc(uni(S.x + (size_t)b * S.cin * xp), valid ? (unsigned)(S.cin * xp) * 4u : 0u);
    const int ub = tb * (4 * W_BT) - S.center;                           // first decimated index needed (>= -5)
    const int q0 = (ub - (ub & 3)) >> 2;                                 // floor(ub / 4)
    const int rowlen = pm ? lp : xl;
    const int q = vt + 96 * item, qa = q0 + q;
    const bool ok = q < W_XQ && qa >= 0 && 4 * qa < rowlen;              // (outside the row: zero padding)
    const int e0 = (chunk * W_CK + 2 * vp) * xp + (pm ? ph * lp : 0) + 4 * qa;
    xq[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (unsigned)e0 * 4u : 0x80000000u, 0, 0);
    xq[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (unsigned)(e0 + xp) * 4u : 0x80000000u, 0, 0);
  };
  auto vl_store = [&](const WSeg& S, int buf, int sub, int item, int slot) {
    const int ub = tb * (4 * W_BT) - S.center;
    const int ua = ub - (ub & 3);                                        // decimated index of slab sample 0
    const int q = vt + 96 * item;
    const int nvalid = pm ? lq + (ph < lr) : (S.xlen > 0 ? S.xlen : len);               // samples of this phase / row
    if (ua + 4 * W_XQ > nvalid) {                                        // last block of the row: zero past the end
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ua + 4 * q + e >= nvalid) { xq[slot][0][e] = 0u; xq[slot][1][e] = 0u; }
    }
    if (q < W_XQ) {
      float* dst = lds + buf * W_SLAB + sub * W_SUB + vp * W_RP2 + 2 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<f32x2*>(dst + e * W_P * 2) = (f32x2){__uint_as_float(xq[slot][0][e]), __uint_as_float(xq[slot][1][e])};
    }
  };
  // A fragments of one step, [mt][half]: half h holds k-steps 4h .. 4h+3
  u32x4 areg[MT][2];
  const int a_lane = (l31 * W_CK + lh * 8) * 4;
  auto load_a_half = [&](int h, const WSeg& S, int chunk, int g, bool valid) {
    const float* up = uni(S.u + ((size_t)((chunk * S.ngrp + g) * 6 + xi) * cout_pad + co0) * W_CK);
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, valid ? W_BM * W_CK * 4 : 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      areg[mt][h] = __builtin_amdgcn_raw_buffer_load_b128(r, a_lane + mt * 32 * W_CK * 4 + 16 * h, 0, 0);
  };

  // BF: the three pieces of one step, [mt][piece]
  u32x4 a3[BF ? MT : 1][3];
  const int a3_lane = (l31 * W_A3 + lh * 4) * 4;
  auto load_a3 = [&](const WSeg& S, int chunk, int g, bool valid) {
    const float* up = uni(S.u + ((size_t)((chunk * S.ngrp + g) * 6 + xi) * cout_pad + co0) * W_A3);
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, valid ? W_BM * W_A3 * 4 : 0);
#pragma unroll
    for (int mt = 0; mt < (BF ? MT : 1); ++mt)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        a3[mt][pc] = __builtin_amdgcn_raw_buffer_load_b128(r, a3_lane + mt * 32 * W_A3 * 4 + 32 * pc, 0, 0);
  };

  // L2 warm-up of the A tiles of the NEXT chunk (all its tap groups, this wave's xi): the register
  // prefetch above is only half a step deep, enough for an L2 hit but not for HBM, and the blocks
  // that share a weight panel run in lockstep, so without this every tile is a first touch for all
  // of them.  One lane per 128-byte line, lanes 0-31 tap group 2j, lanes 32-63 group 2j + 1; the
  // result is never used.  The loads are ordinary builtin loads into two registers that the NEXT prefetch "reads" (an empty asm)
  // before it overwrites them: the compiler then knows when they land.  (Until round 6 this was an inline-asm load into one
  // register the compiler knew nothing about -- correct only while the allocator happened to keep that register for the
  // kernel's whole life: every instantiation that spilled moved it, the late write then hit whatever lived there, and the
  // result was garbage that changed from run to run: tools/exp/ragged_bf_debug.py)
  unsigned pf[2] = {0u, 0u};
  auto prefetch_a = [&](const WSeg& S, int chunk, bool valid) {
    constexpr int ROWF = BF ? W_A3 : W_CK;                                 // floats per weight row and chunk
    const float* up = uni(S.u + ((size_t)(chunk * S.ngrp * 6 + xi) * cout_pad + co0) * ROWF);
    const unsigned gstride = 6u * (unsigned)cout_pad * ROWF * 4u;          // bytes between tap groups
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, valid ? (unsigned)(S.ngrp - 1) * gstride + W_BM * ROWF * 4 : 0u);
    asm volatile("" :: "v"(pf[0]), "v"(pf[1]));          // the previous prefetch has landed before its registers are reused
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // (lanes past the tile's lines: out of range; the 96-byte rows of the BF form are 0.75 W_BM lines, the first
      // 32 lanes' worth of which is touched: enough to start the L2 fill of the tile)
      const unsigned off = l31 < W_BM * ROWF / 32 ? (unsigned)(2 * j + lh) * gstride + (unsigned)l31 * 128u : 0x80000000u;
      pf[j] = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    }
  };

  // ---- prologue -------------------------------------------------------------------------------
  WSeg S0 = load_wseg(&G->seg[0]);
  if constexpr (BF) {
    load_a3(S0, 0, 0, true);
  } else {
    load_a_half(0, S0, 0, 0, true);
    load_a_half(1, S0, 0, 0, true);
  }
  int xbuf = 0;
#pragma unroll
  for (int sub = 0; sub < SUBS; ++sub) {
    if constexpr (VL) {
#pragma unroll
      for (int item = 0; item < W_NITEM; ++item) {
        vl_load(S0, sub, true, item, 0);
        vl_store(S0, 0, sub, item, 0);
      }
    } else {
      load_x(S0, sub, true);
      store_x(0, sub);
    }
  }
  __syncthreads();

  f32x2 c0 = {bc0, bc0}, c1 = {bc1, bc1}, c2 = {bc2, bc2};       // alpha, beta, gamma of this wave's row of B^T
  // The K loop of a chunk is a flat sequence of k-step PAIRS: pair p = 4 g + kp (tap group g,
  // channel pair kp of the chunk), 8 MFMAs each.  One ds_read2st64_b64 fetches one B^T sample for
  // 2 tile columns x 2 k-steps (the wave's second column is 64 tiles = 128 floats on, the next
  // channel pair W_RP2 floats on); the packed math runs over the k-step pair; the samples of pair
  // p + 1 are requested before the MFMAs of pair p, across tap groups, so LDS latency is exposed once
  // per chunk only.
  auto run_segment = [&](auto gc, const WSeg& S, const WSeg& Sn, bool more_seg) {
    constexpr int GC = decltype(gc)::value;
    const int nch = S.cin / W_CK;
    for (int c = 0; c < nch; ++c) {
      const bool last_chunk = c == nch - 1;
      const bool has_next = !last_chunk || more_seg;
      const WSeg& Sx = last_chunk ? Sn : S;              // owner of the next chunk
      const int cx = last_chunk ? 0 : c + 1;
      // the slab rows this chunk occupies are refilled (in the other buffer) by the chunk SUBS ahead
      const int sub = c % SUBS;
      const bool wrap = SUBS == 1 ? last_chunk : c + SUBS >= nch;
      const bool has_slab = SUBS == 1 ? has_next : (!wrap || more_seg);
      const WSeg& Ss = SUBS == 1 ? Sx : (wrap ? Sn : S);
      const int cs = SUBS == 1 ? cx : (wrap ? c + SUBS - nch : c + SUBS);
      const float* xsb = lds + xbuf * W_SLAB + sub * W_SUB + lh * 4 * W_RP2 + (th * 32 + l31) * 2;
      f32x2 xr[2][4][NT];                                // [slot][sample][column] = (k-step 2 kp, 2 kp + 1)
      const int sh = VL ? (tb * (4 * W_BT) - S.center) & 3 : 0;     // slab starts `sh` samples before the first tap
      auto fetch = [&](int slot, int p) {
        const int j0 = 3 * (p >> 2) + sh, kp = p & 3;
        const int o[4] = {bo0, bo1, bo2, bo3};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* q = xsb + (((j0 + o[r]) & 3) * W_P + ((j0 + o[r]) >> 2)) * 2 + kp * W_RP2;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) xr[slot][r][nt] = *reinterpret_cast<const f32x2*>(q + 128 * nt);
        }
      };
      // staging of the NEXT chunk's slab, spread over this chunk's pairs (called with the pair index)
      auto stage = [&](int p) {
        if constexpr (VL) {
          // item 0 at the first pair; with two items per thread the first is stored (the other buffer is
          // free all chunk long) and the second requested a few pairs later, in the same 8 registers
          // (a one-group chunk is too short for that: both items at once, in two register sets)
          if (p == 0) {
            vl_load(Ss, cs, has_slab, 0, 0);
            if (W_NITEM == 2 && GC == 1) vl_load(Ss, cs, has_slab, 1, 1);
            if (th == 0) prefetch_a(Sx, cx, has_next);
          }
          if (W_NITEM == 2 && GC > 1 && p == 4) {
            if (has_slab) vl_store(Ss, xbuf ^ 1, sub, 0, 0);
            vl_load(Ss, cs, has_slab, 1, 0);
          }
          if (W_NITEM == 1 && GC > 1 && p == 4 * (GC - 1) && has_slab) vl_store(Ss, xbuf ^ 1, sub, 0, 0);
        } else {
          if (p == 0) {
            load_x(Ss, cs, has_slab);                     // stored in the last tap group of this chunk
            if (th == 0) prefetch_a(Sx, cx, has_next);
          }
          if (GC > 1 && p == 4 * (GC - 1) && has_slab) store_x(xbuf ^ 1, sub);
        }
      };
      if constexpr (BF) {
        // bf16 x 6 form, one tap group (16-channel k-block) at a time, one tile column at a time: the lane's 8
        // channels of the column -- 4 pairs x 4 samples from the slab, the same packed transform as below -- are
        // split into three bf16 pieces and meet the three weight pieces in 6 MFMAs per 32 x 32 tile.
        const int o[4] = {bo0, bo1, bo2, bo3};
#pragma unroll
        for (int g = 0; g < GC; ++g) {
          stage(4 * g);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            // (one column at a time, also for the instruction scheduler: with both columns' 32 LDS reads hoisted to
            // the top the 64 x 512 tile needs 240 VGPRs more than the 168 a 12-wave block may use)
            // (... and nothing of the next tap group's work moves above this group's MFMAs: the 128 x 256 tile otherwise fills
            // its 168 registers with hoisted reads and spills a hundred)
            __builtin_amdgcn_sched_barrier(0);
            float v[8];
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
              f32x2 x[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int j = 3 * g + sh + o[r];
                x[r] = *reinterpret_cast<const f32x2*>(xsb + ((j & 3) * W_P + (j >> 2)) * 2 + kp * W_RP2 + 128 * nt);
              }
              // (one result per lane on purpose, see split8; same bits as the packed form of the fp32 kernel: every element
              // is one fma chain)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const float pp = __builtin_fmaf(bc0, x[0][e], x[1][e]);
                const float qq = __builtin_fmaf(bc1, x[2][e], x[3][e]);
                v[2 * kp + e] = __builtin_fmaf(bc2, qq, pp);
              }
            }
            bf16x8 bh, bm, bl;
            split8(v, bh, bm, bl);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const bf16x8 ah = __builtin_bit_cast(bf16x8, a3[mt][0]), am = __builtin_bit_cast(bf16x8, a3[mt][1]),
                           al = __builtin_bit_cast(bf16x8, a3[mt][2]);
              f32x16 t = acc[mt][nt];              // small terms first
              t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, t, 0, 0, 0);
              t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, t, 0, 0, 0);
              t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, t, 0, 0, 0);
              t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, t, 0, 0, 0);
              t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, t, 0, 0, 0);
              t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, t, 0, 0, 0);
              acc[mt][nt] = t;
            }
          }
          const bool same_chunk = g + 1 < GC;      // the A registers are free: request the next tap group's pieces
          const WSeg& Sa = same_chunk ? S : Sx;
          load_a3(Sa, same_chunk ? c : cx, same_chunk ? g + 1 : 0, same_chunk || has_next);
        }
      } else {
      fetch(0, 0);
#pragma unroll
      for (int p = 0; p < 4 * GC; ++p) {
        const int g = p >> 2, kp = p & 3, h = kp >> 1;
        stage(p);
        if (p + 1 < 4 * GC) fetch((p + 1) & 1, p + 1);
        f32x2 bf[NT];                              // [column] = B values of k-steps 2 kp, 2 kp + 1
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          // (inline asm keeps the scheduler from interleaving these with the MFMAs below; the hazard recognizer does
          // not look inside asm: VALU result -> MFMA operand needs 2 wait states; the MFMAs read column 0 first, so
          // one s_nop after the last column covers all)
          f32x2 qq;
          asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(bf[nt]) : "s"(c0), "v"(xr[p & 1][0][nt]), "v"(xr[p & 1][1][nt]));
          asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(qq) : "s"(c1), "v"(xr[p & 1][2][nt]), "v"(xr[p & 1][3][nt]));
          if (nt + 1 < NT)
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(bf[nt]) : "s"(c2), "v"(qq));
          else
            asm("v_pk_fma_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(bf[nt]) : "s"(c2), "v"(qq));
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int e = 2 * (kp & 1) + k2;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[mt][h][e]), bf[nt][k2],
                                                                 acc[mt][nt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kp & 1) {                              // this half of the A registers is free: refill it for the next step
          const bool same_chunk = g + 1 < GC;
          const WSeg& Sa = same_chunk ? S : Sx;
          load_a_half(h, Sa, same_chunk ? c : cx, same_chunk ? g + 1 : 0, same_chunk || has_next);
        }
      }
      }
      if constexpr (VL) {
        if (has_slab) {
          if (GC == 1) vl_store(Ss, xbuf ^ 1, sub, 0, 0);
          if (W_NITEM == 2) vl_store(Ss, xbuf ^ 1, sub, 1, GC == 1 ? 1 : 0);
        }
      } else {
        if (GC == 1 && has_slab) store_x(xbuf ^ 1, sub);
      }
      if (has_next && sub == SUBS - 1) {                 // end of a slab: one barrier per SUBS chunks
        __syncthreads();
        xbuf ^= 1;
      }
    }
  };

  // Segments are sorted by tap-group count, descending (host: make_wino_group): one pass over each
  // instantiation instead of a switch inside the segment loop, which costs ~45 VGPRs in spills.
  int sg = 0;
  auto run_all = [&](auto gc) {
    while (sg < nseg && S0.ngrp == decltype(gc)::value) {
      const bool more_seg = sg + 1 < nseg;
      const WSeg Sn = load_wseg(&G->seg[more_seg ? sg + 1 : sg]);
      run_segment(gc, S0, Sn, more_seg);
      S0 = Sn;
      ++sg;
    }
  };
  run_all(std::integral_constant<int, 4>{});
  run_all(std::integral_constant<int, 3>{});
  run_all(std::integral_constant<int, 2>{});
  run_all(std::integral_constant<int, 1>{});

  // ---- epilogue: exchange M_xi through LDS, y = A^T M, bias + residuals, scale, store ----------
  const int nres = uni(G->nres);
  const float scale = G->scale;
  const int cout = uni(G->cout);
  const float* __restrict__ bias = uni(G->bias);
  const int ostride = uni(G->out_stride) > 1 ? uni(G->out_stride) : 1;      // transposed-conv phase: out row = ostride * len
  const int ophase = uni(G->out_phase);
  // out_len > 0: row pitch of out (and res); a transposed conv with an odd (k - u) returns u * len_in + 1 samples per row
  const int opitch = uni(G->out_len) > 0 ? uni(G->out_len) : pitch * ostride;
  const size_t slab = (size_t)b * cout * opitch;
  const unsigned slab_bytes = (unsigned)cout * (unsigned)opitch * 4u;
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(uni((const float*)G->out) + slab, slab_bytes);
  const __amdgpu_buffer_rsrc_t rr0 = make_rsrc(nres > 0 ? uni(G->res[0]) + slab : nullptr, nres > 0 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rr1 = make_rsrc(nres > 1 ? uni(G->res[1]) + slab : nullptr, nres > 1 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rr2 = make_rsrc(nres > 2 ? uni(G->res[2]) + slab : nullptr, nres > 2 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias, bias ? (unsigned)cout * 4u : 0u);
  const bool vec = (pm || (dil == 1 && (len & 3) == 0)) && ostride == 1;   // 4 outputs of a tile = one aligned 16-byte vector
  // Exchange tiles are COLUMN-major, E[th][xi][col (tile) 32][row (co) W_EP]: a lane's 16 accumulators of a
  // 32 x 32 tile are 4 runs of 4 consecutive rows (8 q + 4 lh + 0..3) of its column l31, i.e. four ds_write_b128,
  // and the reader of (column, row quad) gets its six M_xi for 4 rows with six ds_read_b128.  One pass per tile:
  // thread t < 512 = (half th, row quad rq, column col) makes 4 rows x 4 outputs and stores four 16-byte vectors
  // (lanes 0-31 of a store: 512 contiguous bytes of one row).  Against the row-major tiles with 4-byte accesses of
  // round 2 (16 writes + 18 reads per thread and tile in 3 passes, ~2 us per tile with the matrix pipes idle:
  // tools/wino_trace2.py) this is a quarter of the LDS instructions and a third of the address arithmetic.
  const int eth = tid >> 8, erq = (tid >> 5) & 7, ecol = tid & 31;       // reader item (threads 512 .. 767 idle)
  const bool eact = tid < 512;
  // The bias of a thread's rows (all of them now) and the first residual of a sub-tile's 4 rows TWO sub-tiles ahead
  // (the weight registers are free): under load an HBM round trip takes 1.5-2 us, an exchange 1.2 (tools/wino_trace2.py)
  constexpr int kRB = MT * NT < 3 ? MT * NT : MT == 4 ? 2 : 3;      // residual buffers (128-row tile: 2, registers); requests run kRB - 1 ahead
  float bpre[MT][4];
  u32x4 rpre[kRB][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = co0 + mt * 32 + 4 * erq + i;
      bpre[mt][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rbias, (eact && co < cout) ? (unsigned)co * 4u : 0x80000000u, 0, 0));
    }
  auto request = [&](int mt, int nt, u32x4 (&rp)[4]) {
    const int corow = co0 + mt * 32 + 4 * erq;
    const int v0 = tb * (4 * W_BT) + (nt * 64 + eth * 32 + ecol) * 4;
    const bool colok = eact && (v0 + 3) * dil + ph < len;
    const unsigned coloff = (pm ? (unsigned)(ph * lp) : 0u) + (unsigned)v0;
    if (vec && nres > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = colok && corow + i < cout;
        rp[i] = __builtin_amdgcn_raw_buffer_load_b128(
            rr0, ok ? ((unsigned)(corow + i) * (unsigned)opitch + coloff) * 4u : 0x80000000u, 0, 0);
      }
    }
  };
  request(0, 0, rpre[0]);
  if (kRB > 2) request(1 / NT, 1 % NT, rpre[1]);
  __syncthreads();                                   // every wave is out of the K loop: the slab space is free
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      constexpr int kEB = Cfg::EBUFS;
      const int sub = mt * NT + nt;
      float* E = lds + (kEB == 2 ? (sub & 1) * Cfg::EPI : 0);
      if (kEB == 1 && sub > 0) __syncthreads();      // (one buffer: the readers of the previous sub-tile must be done)
      const int corow = co0 + mt * 32 + 4 * erq;                        // + i
      const int v0 = tb * (4 * W_BT) + (nt * 64 + eth * 32 + ecol) * 4;   // decimated index of y[0]
      const bool colok = eact && (v0 + 3) * dil + ph < len;
      if (kRB > 1 && sub + kRB - 1 < MT * NT) request((sub + kRB - 1) / NT, (sub + kRB - 1) % NT, rpre[(sub + kRB - 1) % kRB]);
      {
        float* ew = E + ((th * 6 + xi) * 32 + l31) * W_EP + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(ew + 8 * q) =
              (f32x4){acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
      }
      __syncthreads();
      if (eact) {
        const float* er = E + (eth * 6 * 32 + ecol) * W_EP + 4 * erq;
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(er), m1 = *reinterpret_cast<const f32x4*>(er + 32 * W_EP),
                    m2 = *reinterpret_cast<const f32x4*>(er + 64 * W_EP), m3 = *reinterpret_cast<const f32x4*>(er + 96 * W_EP),
                    m4 = *reinterpret_cast<const f32x4*>(er + 128 * W_EP), m5 = *reinterpret_cast<const f32x4*>(er + 160 * W_EP);
        // (vector index = row i of the quad; the arithmetic per element is the one of every Winograd kernel here)
        const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        const f32x4 y0 = m0 + s12 + s34;
        const f32x4 y1 = __builtin_elementwise_fma((f32x4)(2.f), d34, d12);
        const f32x4 y2 = __builtin_elementwise_fma((f32x4)(4.f), s34, s12);
        const f32x4 y3 = __builtin_elementwise_fma((f32x4)(8.f), d34, d12) + m5;
        // One store path per WAVE, every access an unconditional buffer operation (nothing to do = out-of-range offset):
        // with a per-lane "whole vector?" branch around them the compiler's s_waitcnt in front of each residual use let
        // only 2-3 operations stay in flight, and vmcnt counts loads and stores in order -- every sub-tile then waited
        // for the write acknowledge of the one before it (conv_wino54.hip, tools/exp/w54_fixed_cost.py).
        const float y[4][4] = {{y0[0], y1[0], y2[0], y3[0]}, {y0[1], y1[1], y2[1], y3[1]}, {y0[2], y1[2], y2[2], y3[2]},
                               {y0[3], y1[3], y2[3], y3[3]}};                       // [row i][output q]
        const unsigned phoff = pm ? (unsigned)(ph * lp) : 0u;
        if (vec && __builtin_amdgcn_ballot_w64(!colok) == 0ull) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned off = corow + i < cout ? ((unsigned)(corow + i) * (unsigned)opitch + phoff + (unsigned)v0) * 4u : 0x80000000u;
            const float bv = bpre[mt][i];
            f32x4 o = {y[i][0] + bv, y[i][1] + bv, y[i][2] + bv, y[i][3] + bv};
            if (nres > 0) {
              u32x4 t = rpre[sub % kRB][i];
              f32x4 rs = {__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
              if (nres > 1) {
                t = __builtin_amdgcn_raw_buffer_load_b128(rr1, off, 0, 0);
                rs += (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
              }
              if (nres > 2) {
                t = __builtin_amdgcn_raw_buffer_load_b128(rr2, off, 0, 0);
                rs += (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
              }
              o += rs;
            }
            o *= scale;
            const u32x4 ou = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
            __builtin_amdgcn_raw_buffer_store_b128(ou, ro, off, 0, 0);
          }
        } else {                                        // a row ends inside this wave's tiles, strided or unaligned rows
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int co = corow + i;
            const unsigned rowoff = (unsigned)co * (unsigned)opitch + phoff;
            const float bv = bpre[mt][i];
            unsigned off[4];
            float rs[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int n = ph + dil * (v0 + q);
              off[q] = (co < cout && n < len && (ostride == 1 || n * ostride + ophase < opitch))
                           ? (rowoff + (unsigned)(pm ? v0 + q : n * ostride + ophase)) * 4u : 0x80000000u;
              rs[q] = 0.f;
            }
            if (nres > 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q) rs[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr0, off[q], 0, 0));
              if (nres > 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rs[q] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr1, off[q], 0, 0));
              }
              if (nres > 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rs[q] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr2, off[q], 0, 0));
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float o = y[i][q] + bv;
              if (nres > 0) o += rs[q];
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o * scale), ro, off[q], 0, 0);
            }
          }
        }
      }
    }
  }
  asm volatile("" :: "v"(pf[0]), "v"(pf[1]));            // (the last prefetch: waited for, never used)
}

// out = ((a + b) + c) * scale, 4 elements per thread (the reference's xs += ...; xs / n order)
__global__ __launch_bounds__(256) void mean_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                   const f32x4* __restrict__ c, f32x4* __restrict__ out,
                                                   long long n4, float scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = a[i] + b[i];
  if (c) v += c[i];
  out[i] = v * scale;
}

}  // namespace

extern "C" int fh_mean_f32(const float* a, const float* b, const float* c, float* out, long long n, float scale,
                           void* stream) {
  FH_CHECK_ARG(a && b && out && n > 0 && n % 4 == 0, "fh_mean_f32: bad args (n must be a multiple of 4)");
  FH_CHECK_ARG(((((size_t)a) | ((size_t)b) | ((size_t)c) | ((size_t)out)) & 15) == 0, "fh_mean_f32: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(mean_kernel, dim3(fh_cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)a, (const f32x4*)b, (const f32x4*)c, (f32x4*)out, n / 4, scale);
  FH_CHECK_LAUNCH("fh_mean_f32");
  return FH_OK;
}

// out = (((p0 + p1) + p2) + ...) * scale over up to 12 tensors (split-K partial outputs: fixed order of addition)
namespace {
struct SumArgs {
  const f32x4* p[12];
  int n;
};
__global__ __launch_bounds__(256) void sum_kernel(SumArgs a, f32x4* __restrict__ out, long long n4, float scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = a.p[0][i];
#pragma unroll
  for (int k = 1; k < 12; ++k)
    if (k < a.n) v += a.p[k][i];
  out[i] = v * scale;
}
}  // namespace

extern "C" int fh_sum_f32(const float* const* srcs, int n_srcs, float* out, long long n, float scale, void* stream) {
  FH_CHECK_ARG(srcs && n_srcs >= 1 && n_srcs <= 12 && out && n > 0 && n % 4 == 0,
               "fh_sum_f32: bad args (1..12 sources, n a multiple of 4)");
  SumArgs a;
  a.n = n_srcs;
  for (int k = 0; k < 12; ++k) {
    a.p[k] = (const f32x4*)(k < n_srcs ? srcs[k] : srcs[0]);
    FH_CHECK_ARG(a.p[k] && (((size_t)a.p[k]) & 15) == 0, "fh_sum_f32: source %d is null or not 16-byte aligned", k);
  }
  FH_CHECK_ARG((((size_t)out) & 15) == 0, "fh_sum_f32: out must be 16-byte aligned");
  hipLaunchKernelGGL(sum_kernel, dim3(fh_cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, a, (f32x4*)out, n / 4,
                     scale);
  FH_CHECK_LAUNCH("fh_sum_f32");
  return FH_OK;
}

namespace {
__global__ __launch_bounds__(256) void sum_multi_kernel(const fh_sum_job* __restrict__ jobs) {
  const fh_sum_job& J = jobs[blockIdx.y];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= J.n / 4) return;
  f32x4 v = reinterpret_cast<const f32x4*>(J.src[0])[i];
  for (int k = 1; k < J.n_src; ++k) v += reinterpret_cast<const f32x4*>(J.src[k])[i];
  reinterpret_cast<f32x4*>(J.out)[i] = v * J.scale;
}
}  // namespace

extern "C" int fh_sizeof_sum_job(void) { return (int)sizeof(fh_sum_job); }

extern "C" int fh_sum_multi_f32(const fh_sum_job* jobs, int n_jobs, long long max_n, void* stream) {
  FH_CHECK_ARG(jobs && n_jobs > 0 && n_jobs < 65536 && max_n > 0 && max_n % 4 == 0, "fh_sum_multi_f32: bad args");
  hipLaunchKernelGGL(sum_multi_kernel, dim3(fh_cdiv(max_n / 4, 256), n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
  FH_CHECK_LAUNCH("fh_sum_multi_f32");
  return FH_OK;
}

extern "C" int fh_sizeof_wino_group(void) { return (int)sizeof(fh_wino_group); }

namespace {

template <int MT, int NT, int SUBS, bool VL, bool BF>
int launch_wino_vl(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation,
                int phase_major, hipStream_t stream, const int* run_map, int n_runs, bool xcd_ranges) {
  using Cfg = WCfg<MT, NT, SUBS>;
  FH_CHECK_ARG(cout_pad > 0 && cout_pad % Cfg::BM == 0, "fh_conv_wino_f32: cout_pad %d not a multiple of %d", cout_pad, Cfg::BM);
  const int co_tiles = cout_pad / Cfg::BM;
  const int n_tiles = fh_cdiv(fh_cdiv(len, dilation), 4 * Cfg::BT) * dilation;
  const long long panels = (long long)n_groups * batch * co_tiles;
  const int run_len = fh_cdiv(n_tiles, fh_cdiv(n_tiles, W_RUN));
  const long long runs = run_map ? (long long)n_runs : panels * fh_cdiv(n_tiles, run_len);
  const int rect_r = co_tiles < 32 ? 32 / co_tiles : 1, tpx = fh_cdiv(n_tiles, 8);        // (see the kernel's xcd_ranges branch)
  const long long blocks = xcd_ranges ? 8ll * n_groups * batch * co_tiles * fh_cdiv(tpx, rect_r) * rect_r
                                      : (long long)fh_cdiv(runs, 8) * 8 * run_len;
  FH_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "fh_conv_wino_f32: grid too large");
  const WDivs dv = {fh_make_fastdiv((unsigned)run_len), fh_make_fastdiv((unsigned)fh_cdiv(n_tiles, run_len)),
                    fh_make_fastdiv((unsigned)co_tiles), fh_make_fastdiv((unsigned)batch), fh_make_fastdiv((unsigned)dilation),
                    fh_make_fastdiv((unsigned)rect_r), fh_make_fastdiv((unsigned)(co_tiles * rect_r)),
                    fh_make_fastdiv((unsigned)(co_tiles * fh_cdiv(tpx, rect_r) * rect_r))};
  // > 64 KB of dynamic LDS needs the attribute once per DEVICE (a kernel has one function object per device, and
  // a process may hold models on several): one flag per device ordinal and template instance.
  static std::atomic<bool> lds_opt_in[FH_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FH_MAX_DEVICES) {
    fh_set_error("fh_conv_wino_f32: no current HIP device (or ordinal >= %d)", FH_MAX_DEVICES);
    return FH_E_LAUNCH;
  }
  if (!lds_opt_in[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wino_kernel<MT, NT, SUBS, VL, BF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       Cfg::LDS_FLOATS * 4);
    if (e != hipSuccess) {
      fh_set_error("fh_conv_wino_f32: cannot reserve %d bytes of LDS on device %d: %s", Cfg::LDS_FLOATS * 4, dev, hipGetErrorString(e));
      return FH_E_LAUNCH;
    }
    lds_opt_in[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((conv_wino_kernel<MT, NT, SUBS, VL, BF>), dim3((unsigned)blocks), dim3(W_THREADS), Cfg::LDS_FLOATS * 4,
                     stream, groups, n_groups, batch, co_tiles, n_tiles, run_len, dilation, phase_major, run_map, n_runs,
                     xcd_ranges ? 1 : 0, dv);
  FH_CHECK_LAUNCH("fh_conv_wino_f32");
  return FH_OK;
}

template <int MT, int NT, int SUBS, bool BF>
int launch_wino(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation,
                int phase_major, hipStream_t stream, const int* run_map = nullptr, int n_runs = 0) {
  // 16-byte slab loads need contiguous aligned rows (tensors themselves 16-byte aligned: host plan)
  // (phase_major bit 1: the caller rules the vector loader out -- ragged launches in which some group's rows are
  // not 16-byte aligned; `len` is then only the longest group's length)
  const bool pm = (phase_major & 1) != 0;
  const bool vl = (pm || (dilation == 1 && len % 4 == 0)) && !(phase_major & 2);       // (bit 1 also from FH_WINO_NOVL)
  const bool xr = (phase_major & 4) != 0 && !run_map;          // (bit 2, set by wino_dispatch: FH_WINO_XCD_RANGES)
  return vl ? launch_wino_vl<MT, NT, SUBS, true, BF>(groups, n_groups, batch, cout_pad, len, dilation, pm, stream, run_map, n_runs, xr)
            : launch_wino_vl<MT, NT, SUBS, false, BF>(groups, n_groups, batch, cout_pad, len, dilation, pm, stream, run_map, n_runs, xr);
}

}  // namespace

extern "C" int fh_wino_tile_m(int tile_cfg) {
  tile_cfg &= ~(FH_WINO_BF16X6 | FH_WINO_XCD_RANGES | FH_WINO_NOVL);
  return tile_cfg == 6 ? 128 : tile_cfg == 5 ? 32 : tile_cfg == 4 ? 64 : tile_cfg == 1 ? 96 : tile_cfg == 0 ? 64 : -1;
}

extern "C" int fh_phase_len(int len, int dilation) { return ((len + dilation - 1) / dilation + 3) & ~3; }

namespace {
int wino_dispatch(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation,
                  int phase_major, int tile_cfg, hipStream_t st, const int* run_map, int n_runs) {
  if (tile_cfg & FH_WINO_XCD_RANGES) {
    tile_cfg &= ~FH_WINO_XCD_RANGES;
    phase_major |= 4;
  }
  if (tile_cfg & FH_WINO_NOVL) {
    tile_cfg &= ~FH_WINO_NOVL;
    phase_major |= 2;
  }
  switch (tile_cfg) {
    case 0: return launch_wino<2, 2, 1, false>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
    case 1: return launch_wino<3, 1, 1, false>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
    case 4: return launch_wino<2, 1, 1, false>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
    case 5: return launch_wino<1, 1, 1, false>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
    case 6: return launch_wino<4, 1, 1, false>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
    // + FH_WINO_BF16X6: the groups' weights are three-piece bf16 (pack_wino_weight_bf3), six bf16 MFMAs per k-block
    case FH_WINO_BF16X6 + 0: return launch_wino<2, 2, 1, true>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
    case FH_WINO_BF16X6 + 1: return launch_wino<3, 1, 1, true>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
    case FH_WINO_BF16X6 + 4: return launch_wino<2, 1, 1, true>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
    case FH_WINO_BF16X6 + 5: return launch_wino<1, 1, 1, true>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
    case FH_WINO_BF16X6 + 6: return launch_wino<4, 1, 1, true>(groups, n_groups, batch, cout_pad, len, dilation, phase_major, st, run_map, n_runs);
  }
  fh_set_error("fh_conv_wino_f32: unknown tile_cfg %d", tile_cfg);
  return FH_E_ARG;
}
}  // namespace

extern "C" int fh_conv_wino_f32(const fh_wino_group* groups, int n_groups, int batch, int cout_pad,
                                int len, int dilation, int phase_major, int tile_cfg, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && batch > 0 && len > 0, "fh_conv_wino_f32: bad sizes");
  FH_CHECK_ARG(dilation >= 1 && dilation <= 64, "fh_conv_wino_f32: dilation %d unsupported", dilation);
  // per-clip tensors are addressed with 32-bit byte offsets (buffer descriptors): cin * len * 4 < 2^31
  // is checked by the host plan (flowhigh_amd/vocoder.py) where the shapes are known.
  return wino_dispatch(groups, n_groups, batch, cout_pad, len, dilation, phase_major != 0 ? 1 : 0, tile_cfg, (hipStream_t)stream, nullptr, 0);
}

extern "C" int fh_wino_tile_n(int tile_cfg) { tile_cfg &= ~(FH_WINO_BF16X6 | FH_WINO_XCD_RANGES | FH_WINO_NOVL); return tile_cfg == 0 ? 512 : (fh_wino_tile_m(tile_cfg) > 0 ? 256 : -1); }
extern "C" int fh_wino_run_len(int n_tiles) { return n_tiles > 0 ? fh_cdiv(n_tiles, fh_cdiv(n_tiles, W_RUN)) : -1; }

extern "C" int fh_conv_wino_ragged_f32(const fh_wino_group* groups, int n_groups, int cout_pad, int max_len, int dilation,
                                       int layout_flags, int tile_cfg, const int* run_map, int n_runs, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && max_len > 0 && run_map && n_runs > 0, "fh_conv_wino_ragged_f32: bad sizes");
  FH_CHECK_ARG(layout_flags >= 0 && layout_flags <= 3, "fh_conv_wino_ragged_f32: layout_flags %d (bit 0 phase-major, bit 1 no vector loads)", layout_flags);
  FH_CHECK_ARG(dilation >= 1 && dilation <= 64, "fh_conv_wino_ragged_f32: dilation %d unsupported", dilation);
  return wino_dispatch(groups, n_groups, 1, cout_pad, max_len, dilation, layout_flags, tile_cfg, (hipStream_t)stream, run_map, n_runs);
}
