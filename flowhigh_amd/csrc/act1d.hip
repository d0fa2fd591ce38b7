// Fused anti-aliased periodic activation (HBM-bound kernel).
//
// Replaces Activation1d.forward of BigVGAN
// (/root/reference/src/flowhigh/models/bigvgan/alias_free_torch/act.py:23-28):
//   UpSample1d(2, 12)   resample.py:25-33   replicate pad 5|5, 2 * conv_transpose1d(stride 2), crop 15|15
//   Snake / SnakeBeta   activations.py:48-59,107-120   x + inv_beta * sin^2(alpha x)
//   DownSample1d(2, 12) filter.py:86-95     replicate pad 5|6, conv1d(stride 2)
// The reference runs ~10 aten kernels and materialises the 2x-rate tensor three times per site;
// here the 2x-rate samples only ever exist in LDS: one coalesced read and one coalesced write of
// the [B, C, L] tensor per site.
//
// Closed forms (x index clamped to [0, L-1], z index clamped to [0, 2L-1]; f = 12 taps):
//   z[2i]   = snake( 2 * sum_{q=-3..2} x[i+q] f_up[5-2q] )
//   z[2i+1] = snake( 2 * sum_{q=-2..3} x[i+q] f_up[6-2q] )
//   y[i]    = sum_{k=0..11} z[2i+k-5] f_dn[k]
// The kernel works on the (odd, next even) pairs P_i = (z[2i+1], z[2i+2]): both halves of P_i read the SAME six
// samples x[i-2 .. i+3] (taps (2 f_up[6-2q], 2 f_up[7-2q]), q = -2..3), and y[i] is the sum of both halves of
// sum_{m=0..5} P_{i-3+m} * (f_dn[2m], f_dn[2m+1]): six packed FMAs per pair and per output (the (even, odd) pairing
// of rounds 1-2 straddled both filters: seven each).
//
// Block = 256 threads; a tile = one (group, batch, channel) row segment of TT = 256 PPT - 8 outputs:
//   phase 1: x[t0-8 .. t0+TT+7] -> LDS                         (258 float4, clamped indices)
//   phase 2: thread t makes the 4 consecutive pairs P_i, i = t0 - 4 + 4t .. + 3 -> LDS
//   phase 3: thread t makes the 4 consecutive outputs 4t .. 4t+3 and stores them as one 16-byte vector.
// Every LDS access is a 16-byte vector; the filter and snake arithmetic of phase 2 is on the
// packed-fp32 VALU (v_pk_fma_f32).  (A persistent, software-prefetching variant measured slower.)
#include "fh_common.h"

#include <stdlib.h>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte access at any dword address

// sin^2(a) without libm's sinf, two values at a time.  sin^2 has period pi and is even: Cody-Waite
// reduction by pi in two fused steps (k = rint(a / pi); a - k fl(pi) is exact in one FMA for |a| < 2^15: its bits lie
// between 2^-23 and 2^0; then - k (pi - fl(pi))) to |r| <= pi/2, then
// sin^2(r) = w P(w), w = r^2, P = degree-5 near-minimax fit of sin^2(sqrt w) / w on [0, (pi/2)^2]
// (tools: numpy Chebyshev fit).  No quadrant select.  Absolute error <= 1.6e-7 in fp32 (checked on the
// host against float64), the same size as squaring a 1-ulp sinf.  |a| >= 32768 is patched by the caller.
__device__ __forceinline__ f32x2 sin_squared2(f32x2 a) {
  // k = rint(a / pi) by the 1.5 * 2^23 trick (one packed FMA + one packed add instead of a multiply and two v_rndne)
  const f32x2 k = __builtin_elementwise_fma(a, (f32x2)(0.31830988618379067154f), (f32x2)(12582912.f)) - 12582912.f;
  f32x2 r = __builtin_elementwise_fma(k, (f32x2)(-3.14159274101257324f), a);
  r = __builtin_elementwise_fma(k, (f32x2)(8.742277657347586e-08f), r);
  const f32x2 w = r * r;
  f32x2 p = __builtin_elementwise_fma(w, (f32x2)(-3.6304279547e-06f), (f32x2)(1.3934598246e-04f));
  p = __builtin_elementwise_fma(w, p, (f32x2)(-3.1723924913e-03f));
  p = __builtin_elementwise_fma(w, p, (f32x2)(4.4443175197e-02f));
  p = __builtin_elementwise_fma(w, p, (f32x2)(-3.3333307505e-01f));
  p = __builtin_elementwise_fma(w, p, (f32x2)(1.0f));
  return w * p;
}

__device__ __noinline__ float sin_squared_slow(float a) {   // huge arguments only (never in practice)
  const float s = sinf(a);
  return s * s;
}

#ifndef ACT_PPT_N
#define ACT_PPT_N 4
#endif
constexpr int ACT_PPT = ACT_PPT_N;                   // z pairs (and outputs) per thread, consecutive
#ifndef ACT_THREADS
#define ACT_THREADS 256
#endif
constexpr int ACT_PAIRS = ACT_THREADS * ACT_PPT;     // z pairs of a tile: samples i = t0 - 4 + p
constexpr int ACT_TT = ACT_PAIRS - 8;        // outputs per tile (multiple of 8: tiles start 16-byte aligned)
constexpr int ACT_XS = ACT_TT + 16;          // staged inputs x[t0-8 .. t0+TT+7]
constexpr int ACT_XF4 = ACT_XS / 4;          // ... as float4s

// ---------------------------------------------------------------------------------------------------
// Software-pipelined strips.  A one-tile-per-block form (rounds 1-2) held at most ~1/3 of a CU's loads in flight
// (load -> barrier -> math -> barrier -> math -> store per block, all resident blocks in phase) and stalled at
// ~3 TB/s where a copy reaches 6-7.  Here a block walks ACT_NTILE consecutive tiles
// of the flattened (row, tile) space and requests tile i + 1 before it computes tile i.  All global
// accesses are unconditional buffer operations (invalid = out-of-range offset), so the compiler's
// s_waitcnt are exact: the wait for the prefetched tile does not drain the stores issued after it.
#ifndef ACT_NTILE_N
#define ACT_NTILE_N 4
#endif
constexpr int ACT_NTILE = ACT_NTILE_N;

// PIN / POUT: input / output rows are phase-major for dilation din / dout (see act1d_kernel).
// DIL > 0: the dilation of the phase-major side is this compile-time value (3 and 5 are instantiated: the per-tile
// divisions by it become multiplies); 0: run-time value.
// RAGGED (fh_act1d_ragged_f32): every group is one clip's [C, len_g] tensor (batch 1) with its own length: the row
// geometry (len, phase length, pitch, tiles per row) is then a per-tile, wave-uniform value instead of a launch constant,
// and a block finds its first tile through the groups' tile_base prefix.
template <bool VEC, bool PIN, bool POUT, int DIL = 0, bool RAGGED = false>
__global__ __launch_bounds__(256) void act1d_strip_kernel(const fh_act_group* __restrict__ groups, int batch,
                                                          int channels, int len, int tiles_per_row,
                                                          long long total_tiles, int din_arg, int dout_arg, int n_groups) {
  const int din = (DIL > 0 && PIN) ? DIL : din_arg, dout = (DIL > 0 && POUT) ? DIL : dout_arg;
  static_assert(ACT_THREADS == 256 && ACT_PPT == 4, "strip kernel is written for 256 threads x 4 outputs");
  __shared__ __attribute__((aligned(16))) float xs[ACT_PAIRS + (PIN ? 112 : 16)];   // (PIN: the phase chunks overshoot the tile by < 5 din + 8)
  __shared__ __attribute__((aligned(16))) float zs[2 * ACT_PAIRS + 16];
  __shared__ __attribute__((aligned(16))) float ys[POUT ? ACT_PAIRS : 4];     // outputs of a tile, natural order
  // filter taps of the current group as the pairs the packed FMAs take: [0..5] up (x 2), [6..11] down;
  // rewritten (other buffer) when a tile belongs to another group.  Read as LDS broadcasts: as scalar loads they
  // cost two s_load round trips, ~10 SGPR moves and 8 packed adds per tile.
  __shared__ __attribute__((aligned(16))) f32x2 taps[2][16];
  const int tid = threadIdx.x;
  const long long g0 = (long long)blockIdx.x * ACT_NTILE;
  auto lp_of = [](int l, int d) { return ((l + d - 1) / d + 3) & ~3; };
  // phase-major input: chunk q = (phase p, 4 consecutive decimated samples); <= 2 chunks per thread
  const int nq_in = PIN ? ((ACT_XS + din - 1) / din + 3) / 4 : 1;
  // Outputs stored per tile.  Phase-major output (one length per launch): the largest multiple of 4 dout <= ACT_TT, so
  // that every tile starts on a chunk boundary of every phase: the TT / 4 <= 254 chunks of a tile are whole 16-byte
  // vectors, one per thread, at offsets that are per-thread constants + one uniform term (ragged launches keep ACT_TT
  // and the general chunk geometry below: their tile prefix is computed by the host from fh_act_tile_len()).
  const bool pout_aligned = POUT && !RAGGED;
  const int TT = pout_aligned ? ACT_TT / (4 * dout) * (4 * dout) : ACT_TT;
  const int nq_out = POUT ? (pout_aligned ? TT / (4 * dout) : ((ACT_TT + dout - 1) / dout + 3) / 4) : 1;
  // (phase, chunk-in-phase) of this thread's two chunks: constant over the tiles.  All other divisions by the
  // dilation are done once per tile on wave-uniform values: ceil((x - p) / d) = x / d + (x % d > p), 0 <= p < d.
  int pin_p[2], pin_k[2], pout_p[2], pout_k[2];
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int f = tid + 256 * rep;
    pin_p[rep] = f / nq_in;
    pin_k[rep] = f - pin_p[rep] * nq_in;
    pout_p[rep] = f / nq_out;
    pout_k[rep] = f - pout_p[rep] * nq_out;
  }
  // phase-major input, per tile: tb = t0 - 8 = bu din + rt; the chunk (p, k) starts at u = bu + (rt > p) + 4 k, i.e. at
  // sample t = tb + pin_ct - rt + (rt > p ? din : 0), pin_ct = 4 k din + p (first tile of a row, tb < 0: u = 4 k, t = pin_ct)
  int pin_ct[2];
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) pin_ct[rep] = 4 * pin_k[rep] * din + pin_p[rep];
  struct PinTile { int bu, rt, a, b; };               // uniform per tile
  auto pin_tile = [&](int t0) {
    const int tb = t0 - 8;
    PinTile q;
    if (tb < 0) { q.bu = 0; q.rt = -1; q.a = 8; q.b = 0; }
    else { q.bu = uni(tb / din); q.rt = tb - q.bu * din; q.a = -q.rt; q.b = din; }
    return q;
  };
  // aligned phase-major output: chunk of thread tid = (phase p, chunk k): ys index p + dout (4 k + e), row offset
  // p lp_out + t0 / dout + 4 k floats; t0 / dout = tile (TT / dout)
  const int pout_cy = pout_p[0] < dout ? pout_p[0] + 4 * pout_k[0] * dout : 0x7fffffff;

  struct Tile {                       // wave-uniform description of one flattened tile
    __amdgpu_buffer_rsrc_t rx, ry;
    const fh_act_group* G;
    int t0, tile_in_row, c, gi, len;
    float alpha, inv_beta;            // vector loads, requested one tile ahead together with the tile itself
    f32x2 tap;                        // thread e < 12: tap pair e of the tile's group (loaded only if the group changes)
  };
  // byte offsets of this thread's tap pair inside fh_act_group: up pair j = (f_up[10 - 2j], f_up[11 - 2j]) (the taps of
  // x[i - 2 + j] in z[2i+1] and z[2i+2]), down pair m = (f_dn[2m], f_dn[2m+1])
  unsigned tap_off0, tap_off1;
  float tap_scale;
  {
    const bool up = tid < 6;
    const int j = up ? tid : tid - 6;
    const int i0 = up ? 10 - 2 * j : 2 * j, i1 = i0 + 1;
    const unsigned base = up ? (unsigned)offsetof(fh_act_group, up_taps) : (unsigned)offsetof(fh_act_group, down_taps);
    tap_off0 = tid < 12 ? base + 4u * (unsigned)i0 : 0x80000000u;
    tap_off1 = tid < 12 ? base + 4u * (unsigned)i1 : 0x80000000u;
    tap_scale = up ? 2.f : 1.f;       // (the 2x of UpSample1d folded in: exact)
  }
  // per-thread parts of the tile's load / store byte offsets (+ 4 t0 per tile); 0x80000000 = never in range.  Offsets
  // before the row (t0 = 0) wrap to huge values and offsets past its end exceed the descriptor: both read 0 / drop.
  unsigned ld_off[2];
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int f = tid + 256 * rep;
    ld_off[rep] = f < ACT_XF4 ? (unsigned)((4 * f - 8) * 4) : 0x80000000u;
  }
  const unsigned st_off = ACT_PPT * tid < ACT_TT ? (unsigned)(ACT_PPT * tid * 4) : 0x80000000u;
  // position of a flattened tile: (tile in row, channel, batch, group); the first one of the block is found by
  // 32-bit divisions, the following ones by carrying (a 64-bit division per tile cost ~400 scalar instructions)
  struct Pos { int tile, c, bb, gi, tpr; };       // tpr: tiles per row of group gi
  auto pos_next = [&](Pos p) {
    if (++p.tile == p.tpr) {
      p.tile = 0;
      if (++p.c == channels) {
        p.c = 0;
        if (RAGGED) {
          ++p.gi;
          p.tpr = (uni(groups[p.gi].len) + ACT_TT - 1) / ACT_TT;
        } else if (++p.bb == batch) { p.bb = 0; ++p.gi; }
      }
    }
    return p;
  };
  auto tile_of = [&](const Pos& p, bool ok, int gi_before) {
    Tile T;
    T.c = p.c;
    T.gi = p.gi;
    T.G = groups + p.gi;
    T.len = RAGGED ? uni(T.G->len) : len;
    const int pitch_in = PIN ? din * lp_of(T.len, din) : T.len, pitch_out = POUT ? dout * lp_of(T.len, dout) : T.len;
    const size_t rowi = (size_t)p.bb * channels + p.c;
    T.rx = make_rsrc(uni(T.G->x) + rowi * (size_t)pitch_in, ok ? (unsigned)pitch_in * 4u : 0u);
    T.ry = make_rsrc(uni((const float*)T.G->y) + rowi * (size_t)pitch_out, ok ? (unsigned)pitch_out * 4u : 0u);
    T.t0 = p.tile * TT;
    T.tile_in_row = p.tile;
    // (buffer loads, not flat ones: with a flat load in flight the compiler has to wait with vmcnt(0))
    T.alpha = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
        make_rsrc(uni(T.G->alpha), (unsigned)channels * 4u), (unsigned)T.c * 4u, 0, 0));
    T.inv_beta = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
        make_rsrc(uni(T.G->inv_beta), (unsigned)channels * 4u), (unsigned)T.c * 4u, 0, 0));
    const __amdgpu_buffer_rsrc_t rg = make_rsrc((const float*)T.G, p.gi != gi_before ? (unsigned)sizeof(fh_act_group) : 0u);
    T.tap[0] = tap_scale * __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, tap_off0, 0, 0));
    T.tap[1] = tap_scale * __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, tap_off1, 0, 0));
    return T;
  };
  // x[t0 - 8 + 4 f .. + 3] for f = tid, tid + 256 (258 float4 per tile); out of the row -> 0, patched below
  auto load_tile = [&](const Tile& T, u32x4 (&xr)[2]) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int f = tid + 256 * rep;
      if (PIN) {           // 4 consecutive samples of one phase: 16 bytes at a dword-aligned address
        const PinTile q = pin_tile(T.t0);
        const int p = pin_p[rep];
        const int u = q.bu + (q.rt > p ? 1 : 0) + 4 * pin_k[rep];  // max(0, ceil((t0 - 8 - p) / din)) + 4 k
        const int lp_in = lp_of(T.len, din);
        // (past the row: reads a neighbour phase or falls out of range; such samples are not used)
        xr[rep] = __builtin_amdgcn_raw_buffer_load_b128(T.rx, p < din ? (unsigned)((p * lp_in + u) * 4) : 0x80000000u, 0, 0);
        continue;
      }
      const int t = T.t0 - 8 + 4 * f;
      if (VEC) {
        xr[rep] = __builtin_amdgcn_raw_buffer_load_b128(T.rx, (unsigned)(T.t0 * 4) + ld_off[rep], 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          xr[rep][e] = __builtin_amdgcn_raw_buffer_load_b32(T.rx, f < ACT_XF4 ? (unsigned)((t + e) * 4) : 0x80000000u, 0, 0);
      }
    }
  };

  Pos pos;
  if (RAGGED) {
    int cnt = 0;
    for (int base = 0; base < n_groups; base += 64) {
      const int idx = base + (tid & 63);
      const bool le = idx < n_groups && (long long)groups[idx].tile_base <= g0;
      cnt += __popcll(__ballot(le));
    }
    pos.gi = uni(cnt - 1);
    pos.bb = 0;
    pos.tpr = (uni(groups[pos.gi].len) + ACT_TT - 1) / ACT_TT;
    const unsigned local = (unsigned)g0 - (unsigned)uni(groups[pos.gi].tile_base);
    const unsigned row = local / (unsigned)pos.tpr;
    pos.tile = uni((int)(local - row * (unsigned)pos.tpr));
    pos.c = uni((int)row);
  } else {
    const unsigned g32 = (unsigned)g0;                       // total_tiles < 2^31 (checked by the launcher)
    const unsigned row = g32 / (unsigned)tiles_per_row;
    pos.tile = uni((int)(g32 - row * (unsigned)tiles_per_row));
    const unsigned gb = row / (unsigned)channels;
    pos.c = uni((int)(row - gb * (unsigned)channels));
    pos.gi = uni((int)(gb / (unsigned)batch));
    pos.bb = uni((int)(gb - (unsigned)pos.gi * (unsigned)batch));
    pos.tpr = tiles_per_row;
  }
  Tile T = tile_of(pos, true, -1);
  int tbuf = 0, gi_prev = -1;
  u32x4 cur[2], nxt[2];
  load_tile(T, cur);
#pragma unroll
  for (int it = 0; it < ACT_NTILE; ++it) {
    const bool ok_n = g0 + it + 1 < total_tiles;
    if (ok_n) pos = pos_next(pos);                            // (past the end: stay on the last tile, zero-sized descriptors)
    const Tile Tn = tile_of(pos, ok_n, T.gi);
    const fh_act_group& G = *T.G;
    if (T.gi != gi_prev) {            // first tile of the block or of a group (uniform): publish its taps
      tbuf ^= 1;                      // (other buffer: waves may still be in phase 3 of the previous tile)
      if (tid < 12) taps[tbuf][tid] = T.tap;
    }
    gi_prev = T.gi;
    const f32x2* tp = taps[tbuf];
    const float alpha = T.alpha, inv_beta = T.inv_beta;
    const int len = T.len, zlast = 2 * len - 1;              // (shadow the launch value: this tile's row length)
    const int lp_out = POUT ? lp_of(len, dout) : 0;
    if (PIN) {                 // scatter the phase chunks to their natural positions (stride din, odd: conflict free)
      // Unconditional: a chunk never starts before the tile (t >= tb), what it brings past the tile lands in the slack
      // of xs, and samples past the row end are overwritten by the replicate padding below.
      const PinTile q = pin_tile(T.t0);
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int p = pin_p[rep];
        if (p < din) {
          float* dst = xs + pin_ct[rep] + q.a + (q.rt > p ? q.b : 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e * din] = __uint_as_float(cur[rep][e]);
        }
      }
    } else {
      *reinterpret_cast<u32x4*>(xs + 4 * tid) = cur[0];
      if (tid + 256 < ACT_XF4) *reinterpret_cast<u32x4*>(xs + 4 * (tid + 256)) = cur[1];
    }
    if (it + 1 < ACT_NTILE) load_tile(Tn, nxt);
    __syncthreads();
    const int t0 = T.t0, tb = t0 - 8;
    if (tb < 0 || tb + ACT_XS > len) {             // replicate padding at the row ends (uniform branch, LDS only)
      for (int j = tid; j < ACT_XS; j += 256) {
        const int t = tb + j;
        if (t < 0) xs[j] = xs[-tb];
        else if (t >= len && len - 1 - tb >= 0) xs[j] = xs[len - 1 - tb];
      }
      __syncthreads();
    }
    // phase 2 (as above)
    {
      f32x2 fu2[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) fu2[q] = tp[q];
      float xv[ACT_PPT + 8];
#pragma unroll
      for (int v = 0; v < ACT_PPT / 4 + 2; ++v) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(xs + ACT_PPT * tid + 4 * v);
        xv[4 * v] = t4[0]; xv[4 * v + 1] = t4[1]; xv[4 * v + 2] = t4[2]; xv[4 * v + 3] = t4[3];
      }
      // all 4 pairs through the branch-free path first (4 independent chains for the scheduler), the test for
      // huge arguments once per tile: a branch per pair cost ~12 exec-mask instructions each and kept the
      // chains apart
      f32x2 zout[ACT_PPT], zf[ACT_PPT], arg[ACT_PPT];
#pragma unroll
      for (int r = 0; r < ACT_PPT; ++r) {
        f32x2 z = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 6; ++q) z = __builtin_elementwise_fma((f32x2)(xv[r + 2 + q]), fu2[q], z);
        zf[r] = z;
        arg[r] = z * alpha;
      }
      f32x2 s2[ACT_PPT];
      float amax = 0.f;
#pragma unroll
      for (int r = 0; r < ACT_PPT; ++r) {
        s2[r] = sin_squared2(arg[r]);
        amax = fmaxf(fmaxf(amax, fabsf(arg[r][0])), fabsf(arg[r][1]));     // (v_max3; NaN falls through, as sinf(NaN))
      }
      if (__builtin_expect(amax >= 32768.f, 0)) {
#pragma unroll 1
        for (int r = 0; r < ACT_PPT; ++r)
          if (fabsf(arg[r][0]) >= 32768.f || fabsf(arg[r][1]) >= 32768.f) {
            s2[r][0] = sin_squared_slow(arg[r][0]);
            s2[r][1] = sin_squared_slow(arg[r][1]);
          }
      }
#pragma unroll
      for (int r = 0; r < ACT_PPT; ++r) zout[r] = __builtin_elementwise_fma((f32x2)(inv_beta), s2[r], zf[r]);
      f32x4* zw = reinterpret_cast<f32x4*>(zs + 2 * ACT_PPT * tid);
#pragma unroll
      for (int v = 0; v < ACT_PPT / 2; ++v)
        zw[v] = (f32x4){zout[2 * v][0], zout[2 * v][1], zout[2 * v + 1][0], zout[2 * v + 1][1]};
    }
    __syncthreads();
    // phase 3 (as above); every thread computes, invalid outputs get an out-of-range store offset
    {
      f32x2 fdp[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) fdp[j] = tp[6 + j];
      const int o0 = ACT_PPT * tid;
      const int i0 = t0 + o0;
      float zv[2 * ACT_PPT + 12];        // pairs 4t .. 4t + 9 (the first one is not used: 16-byte reads)
#pragma unroll
      for (int v = 0; v < ACT_PPT / 2 + 3; ++v) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(zs + 2 * ACT_PPT * tid + 4 * v);
        zv[4 * v] = t4[0]; zv[4 * v + 1] = t4[1]; zv[4 * v + 2] = t4[2]; zv[4 * v + 3] = t4[3];
      }
      float out[ACT_PPT];
      const int zbase = 2 * (t0 - 4) + 1;       // zs[m - zbase] = z[m]
#pragma unroll
      for (int r = 0; r < ACT_PPT; ++r) {       // interior form for every output (reads stay inside zs)
        f32x2 a2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 6; ++j)
          a2 = __builtin_elementwise_fma((f32x2){zv[2 * r + 2 + 2 * j], zv[2 * r + 3 + 2 * j]}, fdp[j], a2);
        // (one scalar add per output: left to the vectoriser this became 6 moves + 2 packed adds per tile)
        asm("v_add_f32 %0, %1, %2" : "=v"(out[r]) : "v"(a2[0]), "v"(a2[1]));
      }
      // outputs whose taps leave [0, 2L-1] exist only in the first and the last tile(s) of a row (uniform test)
      if (t0 == 0 || t0 + TT + 3 > len) {
#pragma unroll 1
        for (int r = 0; r < ACT_PPT; ++r) {
          const int i = i0 + r;
          if (!(2 * i - 5 >= 0 && 2 * i + 6 <= zlast) && i < len && o0 < ACT_TT) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
              int m = 2 * i + k - 5;
              m = m < 0 ? 0 : (m > zlast ? zlast : m);
              acc = fmaf(zs[m - zbase], G.down_taps[k], acc);
            }
            out[r] = acc;
          }
        }
      }
      if (POUT && pout_aligned) { // natural order through LDS, then 4 consecutive outputs of one phase per thread
        *reinterpret_cast<f32x4*>(ys + o0) = (f32x4){out[0], out[1], out[2], out[3]};
        __syncthreads();
        // a chunk is stored if its first sample is inside the row (then it is inside the phase's lp_out floats; what
        // it writes past the phase's last sample is padding nobody reads)
        const bool st = pout_cy < len - t0;
        const float* src = ys + (pout_p[0] < dout ? pout_cy : 0);
        u32x4 ou;
#pragma unroll
        for (int e = 0; e < 4; ++e) ou[e] = __float_as_uint(src[e * dout]);
        const int u0 = uni(T.tile_in_row * (TT / dout));
        __builtin_amdgcn_raw_buffer_store_b128(ou, T.ry, st ? (unsigned)((pout_p[0] * lp_out + u0 + 4 * pout_k[0]) * 4) : 0x80000000u, 0, 0);
      } else if (POUT) {         // (ragged launches) the same with the general chunk geometry
        *reinterpret_cast<f32x4*>(ys + o0) = (f32x4){out[0], out[1], out[2], out[3]};
        __syncthreads();
        const int t_end = (t0 + ACT_TT < len ? t0 + ACT_TT : len) - 1;      // last output of this tile
        const int q0 = uni(t0 / dout), r0 = t0 - q0 * dout;
        const int qe = uni(t_end / dout), re = t_end - qe * dout;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const int p = pout_p[rep], k = pout_k[rep];
          const int u = q0 + (r0 > p ? 1 : 0) + 4 * k;                      // ceil((t0 - p) / dout) + 4 k
          const int u_hi = p < dout ? qe - (p > re ? 1 : 0) : -1;           // floor((t_end - p) / dout), -1 if t_end < p
          const unsigned off0 = (unsigned)((p * lp_out + u) * 4);
          if (u + 3 <= u_hi) {
            u32x4 ou;
#pragma unroll
            for (int e = 0; e < 4; ++e) ou[e] = __float_as_uint(ys[(u + e) * dout + p - t0]);
            __builtin_amdgcn_raw_buffer_store_b128(ou, T.ry, off0, 0, 0);
          } else {
            for (int e = 0; e < 4 && u + e <= u_hi; ++e)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ys[(u + e) * dout + p - t0]), T.ry, off0 + 4u * e, 0, 0);
          }
        }
      } else if (VEC) {
        const unsigned off = (unsigned)(t0 * 4) + st_off;       // (len % 4 == 0: a vector is inside the row or past its end)
        const u32x4 ou = {__float_as_uint(out[0]), __float_as_uint(out[1]), __float_as_uint(out[2]), __float_as_uint(out[3])};
        __builtin_amdgcn_raw_buffer_store_b128(ou, T.ry, off, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[r]), T.ry,
                                                (o0 < ACT_TT && i0 + r < len) ? (unsigned)((i0 + r) * 4) : 0x80000000u, 0, 0);
      }
    }
    T = Tn;
    cur[0] = nxt[0];
    cur[1] = nxt[1];
  }
}

// Occupancy cap (fh_act_set_blocks_per_cu).  At its natural 7 blocks (28 waves) per CU this kernel keeps the vector ALUs, the
// LDS and HBM busy at once, and on some MI355X boxes the power management answers with a lower shader clock that outlasts the
// NEXT launch: a Winograd launch that follows an activation launch then runs at 2.11 instead of 2.38 GHz (+14 % time), far more
// than the activation launch itself costs (tools/clock_dip_probe.py; a device copy of the same bytes does not do it).  With 3
// blocks per CU most of the dip is gone (a step is 5-7 % faster on such a box); on boxes without the dip the cap costs the
// activation 25 %.  So it is a per-device run-time setting that the host calibrates (vocoder.calibrate_act_occupancy): the
// launchers ask for UNUSED dynamic LDS so that only the wanted number of blocks fits the CU's 160 KB.
std::atomic<int> g_act_blocks_per_cu[FH_MAX_DEVICES];

unsigned act_pad_lds_bytes() {          // static LDS of the variants: 12.7-17.2 KB; total must lie in (160 KB / (n + 1), 160 KB / n]
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FH_MAX_DEVICES) return 0u;
  switch (g_act_blocks_per_cu[dev].load(std::memory_order_relaxed)) {
    // (n = 5: 26.67 KB < static + pad <= 32 KB for EVERY variant: 12.7 + 14.5 = 27.2, 17.2 + 14.5 = 31.7; 15 KB put the largest
    // variant at 32.2 KB = 4 blocks)
    case 5: return 14u * 1024u + 512u;
    case 4: return 22u * 1024u;
    case 3: return 36u * 1024u;
    case 2: return 44u * 1024u;
    default: return 0u;
  }
}

}  // namespace

extern "C" int fh_act_set_blocks_per_cu(int blocks) {
  FH_CHECK_ARG(blocks == 0 || (blocks >= 2 && blocks <= 5), "fh_act_set_blocks_per_cu: %d (0 = no cap, or 2..5)", blocks);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FH_MAX_DEVICES) {
    fh_set_error("fh_act_set_blocks_per_cu: no current HIP device (or ordinal >= %d)", FH_MAX_DEVICES);
    return FH_E_LAUNCH;
  }
  g_act_blocks_per_cu[dev].store(blocks, std::memory_order_relaxed);
  return FH_OK;
}

extern "C" int fh_act_get_blocks_per_cu(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FH_MAX_DEVICES) return 0;
  return g_act_blocks_per_cu[dev].load(std::memory_order_relaxed);
}

extern "C" int fh_sizeof_act_group(void) { return (int)sizeof(fh_act_group); }

extern "C" int fh_act1d_grouped_pm_f32(const fh_act_group* groups, int n_groups, int batch,
                                       int channels, int len, int din, int dout, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && batch > 0 && channels > 0 && len > 0, "fh_act1d_grouped_f32: bad sizes");
  FH_CHECK_ARG(din >= 1 && dout >= 1, "fh_act1d_grouped_pm_f32: bad dilations %d / %d", din, dout);
  // (phase-major output: a tile stores the largest multiple of 4 dout outputs <= ACT_TT, see the kernel)
  const int tiles = fh_cdiv(len, dout > 1 && dout <= 16 ? ACT_TT / (4 * dout) * (4 * dout) : ACT_TT);
  const long long blocks = (long long)n_groups * batch * channels * tiles;
  FH_CHECK_ARG(blocks < (1ll << 31), "fh_act1d_grouped_f32: grid too large");
  // chunks of a tile in phase-major form must fit 2 per thread: ceil(ceil(XS / d) / 4) * d <= 512; 32-bit row offsets
  FH_CHECK_ARG(din <= 16 && dout <= 16, "fh_act1d_grouped_pm_f32: phase-major dilations above 16 are not supported (%d / %d)", din, dout);
  FH_CHECK_ARG((long long)len * 4 * (din > dout ? din : dout) < (1ll << 31), "fh_act1d_grouped_f32: rows of %d samples are too long", len);
  {
    const long long strips = (blocks + ACT_NTILE - 1) / ACT_NTILE;
    const bool vec = (len & 3) == 0;   // rows 16-byte aligned provided the tensors are (checked by the host plan)
    const unsigned pad = act_pad_lds_bytes();
#define FH_ACT_LAUNCH(V, PI, PO)                                                                              \
  hipLaunchKernelGGL((act1d_strip_kernel<V, PI, PO>), dim3((unsigned)strips), dim3(256), pad, (hipStream_t)stream, \
                     groups, batch, channels, len, tiles, blocks, din, dout, n_groups)
#define FH_ACT_LAUNCH_D(PI, PO, D)                                                                            \
  hipLaunchKernelGGL((act1d_strip_kernel<true, PI, PO, D>), dim3((unsigned)strips), dim3(256), pad, (hipStream_t)stream, \
                     groups, batch, channels, len, tiles, blocks, din, dout, n_groups)
    if (din > 1 && dout > 1) FH_ACT_LAUNCH(false, true, true);
    else if (din > 1) {
      if (vec && din == 3) FH_ACT_LAUNCH_D(true, false, 3);
      else if (vec && din == 5) FH_ACT_LAUNCH_D(true, false, 5);
      else if (vec) FH_ACT_LAUNCH(true, true, false);
      else FH_ACT_LAUNCH(false, true, false);
    } else if (dout > 1) {
      if (vec && dout == 3) FH_ACT_LAUNCH_D(false, true, 3);
      else if (vec && dout == 5) FH_ACT_LAUNCH_D(false, true, 5);
      else if (vec) FH_ACT_LAUNCH(true, false, true);
      else FH_ACT_LAUNCH(false, false, true);
    } else { if (vec) FH_ACT_LAUNCH(true, false, false); else FH_ACT_LAUNCH(false, false, false); }
#undef FH_ACT_LAUNCH_D
#undef FH_ACT_LAUNCH
    FH_CHECK_LAUNCH("fh_act1d_grouped_f32");
  }
  return FH_OK;
}

extern "C" int fh_act_tile_len(void) { return ACT_TT; }

extern "C" int fh_act1d_ragged_f32(const fh_act_group* groups, int n_groups, int channels, int din, int dout,
                                   long long total_tiles, int all_len_mult4, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && channels > 0 && total_tiles > 0 && total_tiles < (1ll << 31),
               "fh_act1d_ragged_f32: bad sizes");
  FH_CHECK_ARG(din >= 1 && dout >= 1 && din <= 16 && dout <= 16, "fh_act1d_ragged_f32: bad dilations %d / %d", din, dout);
  {                                                                  // (row bytes < 2^31: checked by the host plan)
    const long long strips = (total_tiles + ACT_NTILE - 1) / ACT_NTILE;
    const bool vec = all_len_mult4 != 0;
    const unsigned pad = act_pad_lds_bytes();
#define FH_ACT_RLAUNCH(V, PI, PO)                                                                                       \
  hipLaunchKernelGGL((act1d_strip_kernel<V, PI, PO, 0, true>), dim3((unsigned)strips), dim3(256), pad, (hipStream_t)stream, \
                     groups, 1, channels, 0, 1, total_tiles, din, dout, n_groups)
    if (din > 1 && dout > 1) FH_ACT_RLAUNCH(false, true, true);
    else if (din > 1) { if (vec) FH_ACT_RLAUNCH(true, true, false); else FH_ACT_RLAUNCH(false, true, false); }
    else if (dout > 1) { if (vec) FH_ACT_RLAUNCH(true, false, true); else FH_ACT_RLAUNCH(false, false, true); }
    else { if (vec) FH_ACT_RLAUNCH(true, false, false); else FH_ACT_RLAUNCH(false, false, false); }
#undef FH_ACT_RLAUNCH
    FH_CHECK_LAUNCH("fh_act1d_ragged_f32");
  }
  return FH_OK;
}

extern "C" int fh_act1d_grouped_f32(const fh_act_group* groups, int n_groups, int batch,
                                    int channels, int len, void* stream) {
  return fh_act1d_grouped_pm_f32(groups, n_groups, batch, channels, len, 1, 1, stream);
}
