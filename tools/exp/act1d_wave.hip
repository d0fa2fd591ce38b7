// EXPERIMENT (round 3, not in the product library): flowhigh_amd/csrc/act1d.hip + act1d_wave_kernel, a barrier-free form
// of the fused activation in which every WAVE owns its tile (neighbour samples by DPP wave shifts, phase-major rows
// through a wave-private LDS transpose).  Bit-identical to the strip kernel, all activation tests pass, and it is NOT
// faster: plain rows 64.9-68.5 us against 61.1-64.3 us, phase-major rows 15-20 % slower (profiles/r03_act_wave.txt) --
// the block barriers are not what the strip kernel waits for.
//   tools/build_variant.sh actw tools/exp/act1d_wave.hip=act1d.hip
//   FH_LIB_PATH=flowhigh_amd/lib/abl/actw.so python tools/act_bench.py      (FH_ACT_STRIP=1: the strip kernel)
//
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
//
// Block = 256 threads, one (group, batch, channel) row segment of TT = 256 PPT - 8 outputs:
//   phase 1: x[t0-8 .. t0+TT+7] -> LDS                         (258 float4, clamped indices)
//   phase 2: thread t makes the 4 consecutive (even, odd) pairs 4t .. 4t+3 of z -> LDS
//   phase 3: thread t makes the 4 consecutive outputs 4t .. 4t+3 and stores them as one 16-byte vector.
// Every LDS access is a 16-byte vector; the filter and snake arithmetic of phase 2 is on the
// packed-fp32 VALU (v_pk_fma_f32).  (A persistent, software-prefetching variant measured slower.)
#include "fh_common.h"

#include <stdlib.h>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte access at any dword address

// sin^2(a) without libm's sinf, two values at a time.  sin^2 has period pi and is even: Cody-Waite
// reduction by pi in three fused steps (exact products for |k| < 2^15) to |r| <= pi/2, then
// sin^2(r) = w P(w), w = r^2, P = degree-5 near-minimax fit of sin^2(sqrt w) / w on [0, (pi/2)^2]
// (tools: numpy Chebyshev fit).  No quadrant select.  Absolute error <= 1.6e-7 in fp32 (checked on the
// host against float64), the same size as squaring a 1-ulp sinf.  |a| >= 32768 is patched by the caller.
__device__ __forceinline__ f32x2 sin_squared2(f32x2 a) {
  const f32x2 t = a * 0.31830988618379067154f;
  const f32x2 k = {rintf(t[0]), rintf(t[1])};
  f32x2 r = __builtin_elementwise_fma(k, (f32x2)(-3.140625f), a);
  r = __builtin_elementwise_fma(k, (f32x2)(-9.67502593994140625e-4f), r);
  r = __builtin_elementwise_fma(k, (f32x2)(-1.5099579897537296e-7f), r);
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

// One block = one (group, batch, channel, tile).  Thread t owns pairs 4t .. 4t+3 and outputs
// 4t .. 4t+3: every LDS access is a 16-byte vector, and when rows are 16-byte aligned
// (len % 4 == 0) so is every global access -- 4-byte-per-lane loads ran this kernel at 2.5 TB/s.
// din / dout > 1: the input / output tensor is phase-major for that dilation (fh_phase_len, include/
// flowhigh_hip.h): element t of a row lives at (t % d) * lp + t / d.  A dilated Winograd conv between two
// such launches then works on contiguous runs.  Consecutive lanes still touch consecutive t, i.e. d runs of
// 64 / d contiguous floats per wave instruction.
// RAGGED (fh_act1d_ragged_f32): every group is one clip's [C, len_g] tensor with its own length (fh_act_group.len);
// block -> (group, channel, tile) through the groups' tile_base prefix (ascending; one ballot per 64 groups).
template <bool RAGGED>
__global__ __launch_bounds__(ACT_THREADS) void act1d_kernel(const fh_act_group* __restrict__ groups,
                                                    int batch, int channels, int len,
                                                    int tiles_per_row, int din, int dout, int n_groups) {
  __shared__ __attribute__((aligned(16))) float xs[ACT_PAIRS + 16];
  __shared__ __attribute__((aligned(16))) float zs[2 * ACT_PAIRS + 16];

  int gsel = 0, local = blockIdx.x;
  if (RAGGED) {
    int cnt = 0;
    for (int base = 0; base < n_groups; base += 64) {
      const int idx = base + (int)(threadIdx.x & 63);
      const bool le = idx < n_groups && groups[idx].tile_base <= (int)blockIdx.x;
      cnt += __popcll(__ballot(le));
    }
    gsel = __builtin_amdgcn_readfirstlane(cnt - 1);
    len = __builtin_amdgcn_readfirstlane(groups[gsel].len);
    tiles_per_row = (len + ACT_TT - 1) / ACT_TT;
    local = (int)blockIdx.x - __builtin_amdgcn_readfirstlane(groups[gsel].tile_base);
  }
  const int tile = local % tiles_per_row;
  const int row = local / tiles_per_row;          // (g * batch + b) * channels + c
  const int c = row % channels;
  const int gb = RAGGED ? gsel : row / channels;
  const fh_act_group& G = groups[RAGGED ? gsel : gb / batch];
  const int b = RAGGED ? 0 : gb % batch;
  const int lp_in = ((len + din - 1) / din + 3) & ~3, lp_out = ((len + dout - 1) / dout + 3) & ~3;
  const size_t rowi = (size_t)b * channels + c;
  const float* __restrict__ x = G.x + rowi * (din > 1 ? (size_t)din * lp_in : (size_t)len);
  float* __restrict__ y = G.y + rowi * (dout > 1 ? (size_t)dout * lp_out : (size_t)len);
  const float alpha = G.alpha[c];
  const float inv_beta = G.inv_beta[c];
  const int t0 = tile * ACT_TT;
  const int tid = threadIdx.x;
  const int zlast = 2 * len - 1;
  const bool vec = (len & 3) == 0 && ((((size_t)G.x) | ((size_t)G.y)) & 15) == 0;

  // phase 1: xs[j] = x[clamp(t0 - 8 + j)], j < XS; thread t stages float4 #t, #256+t, ...
  if (din > 1) {
    const int tb = t0 - 8;
    if (tb >= 0 && tb + ACT_XS <= len) {
      // tile interior: every thread fetches 4 consecutive samples of ONE phase (16 bytes, contiguous in the
      // phase-major row) and scatters them to their natural positions in LDS (stride din, odd -> conflict free)
      const int nq = ((ACT_XS + din - 1) / din + 3) / 4;
      for (int q = tid; q < din * nq; q += ACT_THREADS) {
        const int p = q / nq, k = q - p * nq;
        const int u_lo = (tb - p + din - 1) / din;
        const int u_hi = (tb + ACT_XS - 1 - p) / din;
        const int u = u_lo + 4 * k;
        const float* src = x + p * lp_in + u;
        if (u + 3 <= u_hi) {
          const f32x4u v = *reinterpret_cast<const f32x4u*>(src);
#pragma unroll
          for (int e = 0; e < 4; ++e) xs[(u + e) * din + p - tb] = v[e];
        } else {
          for (int e = 0; e < 4 && u + e <= u_hi; ++e) xs[(u + e) * din + p - tb] = src[e];
        }
      }
    } else {
      for (int j = tid; j < ACT_XS; j += ACT_THREADS) {
        int t = tb + j;
        t = t < 0 ? 0 : (t > len - 1 ? len - 1 : t);
        const int u = t / din;
        xs[j] = x[(t - u * din) * lp_in + u];
      }
    }
  } else
#pragma unroll
  for (int rep = 0; rep < (ACT_XF4 + ACT_THREADS - 1) / ACT_THREADS; ++rep) {
    const int f = tid + ACT_THREADS * rep;
    if (f >= ACT_XF4) break;
    const int t = t0 - 8 + 4 * f;
    f32x4 v;
    if (vec && t >= 0 && t + 3 < len) {
      v = *reinterpret_cast<const f32x4*>(x + t);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int tt = t + e;
        tt = tt < 0 ? 0 : (tt > len - 1 ? len - 1 : tt);
        v[e] = x[tt];
      }
    }
    *reinterpret_cast<f32x4*>(xs + 4 * f) = v;
  }
  __syncthreads();

  // phase 2: pairs p = PPT tid + r, sample i = t0 - 4 + p; x[i+q] is xs[p + q + 4], q in [-3, 3]
  f32x2 fu2[7];             // taps of x[i-3 .. i+3] for (z[2i], z[2i+1]); the unused end tap is 0
#pragma unroll
  for (int q = -3; q <= 3; ++q) {
    fu2[q + 3][0] = q <= 2 ? 2.f * G.up_taps[5 - 2 * q] : 0.f;      // (the 2x of UpSample1d folded in: exact)
    fu2[q + 3][1] = q >= -2 ? 2.f * G.up_taps[6 - 2 * q] : 0.f;
  }
  {
    float xv[ACT_PPT + 8];  // xs[PPT tid .. PPT tid + PPT + 7]; pair r uses xv[r + 1 .. r + 7]
#pragma unroll
    for (int v = 0; v < ACT_PPT / 4 + 2; ++v) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(xs + ACT_PPT * tid + 4 * v);
      xv[4 * v] = t4[0]; xv[4 * v + 1] = t4[1]; xv[4 * v + 2] = t4[2]; xv[4 * v + 3] = t4[3];
    }
    f32x2 zout[ACT_PPT];
#pragma unroll
    for (int r = 0; r < ACT_PPT; ++r) {
      f32x2 z = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 7; ++q) z = __builtin_elementwise_fma((f32x2)(xv[r + 1 + q]), fu2[q], z);
      const f32x2 arg = z * alpha;
      f32x2 s2 = sin_squared2(arg);
      if (__builtin_expect(fabsf(arg[0]) >= 32768.f || fabsf(arg[1]) >= 32768.f, 0)) {
        s2[0] = sin_squared_slow(arg[0]);
        s2[1] = sin_squared_slow(arg[1]);
      }
      zout[r] = __builtin_elementwise_fma((f32x2)(inv_beta), s2, z);
    }
    // positions outside [0, 2L-1] are never used directly: phase 3 clamps its index instead
    f32x4* zw = reinterpret_cast<f32x4*>(zs + 2 * ACT_PPT * tid);
#pragma unroll
    for (int v = 0; v < ACT_PPT / 2; ++v)
      zw[v] = (f32x4){zout[2 * v][0], zout[2 * v][1], zout[2 * v + 1][0], zout[2 * v + 1][1]};
  }
  __syncthreads();

  // phase 3: outputs o = 4 tid + r (i = t0 + o); z[m] is zs[m - 2 (t0 - 4)], so
  //          y[i] = sum_k zs[2 o + 3 + k] f_dn[k]  (interior) -- taps on zs[8 tid + 2 r + 3 ..]
  float fd[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) fd[k] = G.down_taps[k];
  // taps as aligned (even, odd) z pairs: pair j of output r is (zv[2r + 2 + 2j], zv[2r + 3 + 2j]) = taps (2j - 1, 2j)
  f32x2 fdp[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    fdp[j][0] = j > 0 ? fd[2 * j - 1] : 0.f;
    fdp[j][1] = j < 6 ? fd[2 * j] : 0.f;
  }
  const int o0 = ACT_PPT * tid;
  const int i0 = t0 + o0;
  if (dout == 1 && (o0 >= ACT_TT || i0 >= len)) return;
  float zv[2 * ACT_PPT + 16];   // zs[2 PPT tid .. + 2 PPT + 15]
#pragma unroll
  for (int v = 0; v < ACT_PPT / 2 + 4; ++v) {
    const f32x4 t4 = *reinterpret_cast<const f32x4*>(zs + 2 * ACT_PPT * tid + 4 * v);
    zv[4 * v] = t4[0]; zv[4 * v + 1] = t4[1]; zv[4 * v + 2] = t4[2]; zv[4 * v + 3] = t4[3];
  }
  float out[ACT_PPT];
  const int zbase = 2 * (t0 - 4);
#pragma unroll
  for (int r = 0; r < ACT_PPT; ++r) {
    const int i = i0 + r;
    float acc = 0.f;
    if (2 * i - 5 >= 0 && 2 * i + 6 <= zlast) {
      f32x2 a2 = {0.f, 0.f};                          // even-tap and odd-tap partial sums, packed
#pragma unroll
      for (int j = 0; j < 7; ++j)
        a2 = __builtin_elementwise_fma((f32x2){zv[2 * r + 2 + 2 * j], zv[2 * r + 3 + 2 * j]}, fdp[j], a2);
      acc = a2[0] + a2[1];
    } else if (i < len) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        int m = 2 * i + k - 5;
        m = m < 0 ? 0 : (m > zlast ? zlast : m);
        acc = fmaf(zs[m - zbase], fd[k], acc);
      }
    }
    out[r] = acc;
  }
  if (dout > 1) {          // through LDS (xs is free now) so that every store instruction writes runs
#pragma unroll
    for (int v = 0; v < ACT_PPT / 4; ++v)
      *reinterpret_cast<f32x4*>(xs + o0 + 4 * v) = (f32x4){out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
    __syncthreads();
    const int t_end = (t0 + ACT_TT < len ? t0 + ACT_TT : len) - 1;      // last output of this tile
    const int nq = ((ACT_TT + dout - 1) / dout + 3) / 4;
    for (int q = tid; q < dout * nq; q += ACT_THREADS) {                         // 4 consecutive outputs of one phase
      const int p = q / nq, k = q - p * nq;
      const int u_lo = (t0 - p + dout - 1) / dout;
      const int u_hi = t_end >= p ? (t_end - p) / dout : -1;
      const int u = u_lo + 4 * k;
      float* dst = y + p * lp_out + u;
      if (u + 3 <= u_hi) {
        f32x4u v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = xs[(u + e) * dout + p - t0];
        *reinterpret_cast<f32x4u*>(dst) = v;
      } else {
        for (int e = 0; e < 4 && u + e <= u_hi; ++e) dst[e] = xs[(u + e) * dout + p - t0];
      }
    }
    return;
  }
#pragma unroll
  for (int v = 0; v < ACT_PPT / 4; ++v) {   // o0 + PPT - 1 < TT always holds for o0 < TT (TT % PPT == 0)
    if (vec && i0 + 4 * v + 3 < len) {
      *reinterpret_cast<f32x4*>(y + i0 + 4 * v) = (f32x4){out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
    } else {
#pragma unroll
      for (int r = 4 * v; r < 4 * v + 4; ++r)
        if (i0 + r < len) y[i0 + r] = out[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Plain-layout launches: software-pipelined strips.  The kernel above holds at most ~1/3 of a CU's
// loads in flight (load -> barrier -> math -> barrier -> math -> store per block, all resident blocks in
// phase) and stalls at ~3 TB/s where a copy reaches 6-7.  Here a block walks ACT_NTILE consecutive tiles
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
  __shared__ __attribute__((aligned(16))) float xs[ACT_PAIRS + 16];
  __shared__ __attribute__((aligned(16))) float zs[2 * ACT_PAIRS + 16];
  __shared__ __attribute__((aligned(16))) float ys[POUT ? ACT_PAIRS : 4];     // outputs of a tile, natural order
  // filter taps of the current group as the (even, odd) pairs the packed FMAs take: [0..6] up (x 2), [7..13] down;
  // rewritten (other buffer) when a tile belongs to another group.  Read as LDS broadcasts: as scalar loads they
  // cost two s_load round trips, ~10 SGPR moves and 8 packed adds per tile.
  __shared__ __attribute__((aligned(16))) f32x2 taps[2][16];
  const int tid = threadIdx.x;
  const long long g0 = (long long)blockIdx.x * ACT_NTILE;
  auto lp_of = [](int l, int d) { return ((l + d - 1) / d + 3) & ~3; };
  // phase-major input: chunk q = (phase p, 4 consecutive decimated samples); <= 2 chunks per thread
  const int nq_in = PIN ? ((ACT_XS + din - 1) / din + 3) / 4 : 1;
  const int nq_out = POUT ? ((ACT_TT + dout - 1) / dout + 3) / 4 : 1;
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

  struct Tile {                       // wave-uniform description of one flattened tile
    __amdgpu_buffer_rsrc_t rx, ry;
    const fh_act_group* G;
    int t0, c, gi, len;
    float alpha, inv_beta;            // vector loads, requested one tile ahead together with the tile itself
    f32x2 tap;                        // thread e < 14: tap pair e of the tile's group (loaded only if the group changes)
  };
  // byte offsets of this thread's tap pair inside fh_act_group (out of range = 0: the unused end taps)
  unsigned tap_off0, tap_off1;
  float tap_scale;
  {
    const bool up = tid < 7;
    const int j = up ? tid : tid - 7;
    const int i0 = up ? 11 - 2 * j : 2 * j - 1, i1 = up ? 12 - 2 * j : 2 * j;
    const bool v0 = tid < 14 && (up ? j <= 5 : j > 0), v1 = tid < 14 && (up ? j >= 1 : j < 6);
    const unsigned base = up ? (unsigned)offsetof(fh_act_group, up_taps) : (unsigned)offsetof(fh_act_group, down_taps);
    tap_off0 = v0 ? base + 4u * (unsigned)i0 : 0x80000000u;
    tap_off1 = v1 ? base + 4u * (unsigned)i1 : 0x80000000u;
    tap_scale = up ? 2.f : 1.f;       // (the 2x of UpSample1d folded in: exact)
  }
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
    T.t0 = p.tile * ACT_TT;
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
        const int tb8 = uni(T.t0 - 8 + 8 * din);                   // >= 0
        const int qt = uni(tb8 / din), rt = tb8 - qt * din;
        const int p = pin_p[rep], k = pin_k[rep];
        const int ul = qt - 8 + (rt > p ? 1 : 0);                  // ceil((t0 - 8 - p) / din)
        const int u = (ul > 0 ? ul : 0) + 4 * k;
        const int lp_in = lp_of(T.len, din);
        // (past the row: reads a neighbour phase or falls out of range; such samples are not used)
        xr[rep] = __builtin_amdgcn_raw_buffer_load_b128(T.rx, p < din ? (unsigned)((p * lp_in + u) * 4) : 0x80000000u, 0, 0);
        continue;
      }
      const int t = T.t0 - 8 + 4 * f;
      if (VEC) {
        xr[rep] = __builtin_amdgcn_raw_buffer_load_b128(T.rx, f < ACT_XF4 ? (unsigned)(t * 4) : 0x80000000u, 0, 0);
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
      if (tid < 14) taps[tbuf][tid] = T.tap;
    }
    gi_prev = T.gi;
    const f32x2* tp = taps[tbuf];
    const float alpha = T.alpha, inv_beta = T.inv_beta;
    const int len = T.len, zlast = 2 * len - 1;              // (shadow the launch value: this tile's row length)
    const int lp_out = POUT ? lp_of(len, dout) : 0;
    if (PIN) {                 // scatter the phase chunks to their natural positions (stride din, odd: conflict free)
      const int tb = T.t0 - 8;
      const int tb8 = uni(tb + 8 * din);
      const int qt = uni(tb8 / din), rt = tb8 - qt * din;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int p = pin_p[rep], k = pin_k[rep];
        const int ul = qt - 8 + (rt > p ? 1 : 0);
        const int u = (ul > 0 ? ul : 0) + 4 * k;
        if (p < din) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int t = (u + e) * din + p;
            if (t >= 0 && t < len && t - tb < ACT_XS) xs[t - tb] = __uint_as_float(cur[rep][e]);
          }
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
      f32x2 fu2[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) fu2[q] = tp[q];
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
        for (int q = 0; q < 7; ++q) z = __builtin_elementwise_fma((f32x2)(xv[r + 1 + q]), fu2[q], z);
        zf[r] = z;
        arg[r] = z * alpha;
      }
#if defined(ACT_ABL) && (ACT_ABL & 2)
#pragma unroll
      for (int r = 0; r < ACT_PPT; ++r) zout[r] = (f32x2){xv[r + 1], xv[r + 2]} * alpha + inv_beta;
#else
      f32x2 s2[ACT_PPT];
      float amax = 0.f;
#pragma unroll
      for (int r = 0; r < ACT_PPT; ++r) {
#if defined(ACT_ABL) && (ACT_ABL & 1)
        s2[r] = arg[r];
#else
        s2[r] = sin_squared2(arg[r]);
        amax = fmaxf(amax, fmaxf(fabsf(arg[r][0]), fabsf(arg[r][1])));     // (NaN falls through, as sinf(NaN))
#endif
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
#endif
      f32x4* zw = reinterpret_cast<f32x4*>(zs + 2 * ACT_PPT * tid);
#pragma unroll
      for (int v = 0; v < ACT_PPT / 2; ++v)
        zw[v] = (f32x4){zout[2 * v][0], zout[2 * v][1], zout[2 * v + 1][0], zout[2 * v + 1][1]};
    }
    __syncthreads();
    // phase 3 (as above); every thread computes, invalid outputs get an out-of-range store offset
    {
      f32x2 fdp[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) fdp[j] = tp[7 + j];
      const int o0 = ACT_PPT * tid;
      const int i0 = t0 + o0;
      float zv[2 * ACT_PPT + 16];
#pragma unroll
      for (int v = 0; v < ACT_PPT / 2 + 4; ++v) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(zs + 2 * ACT_PPT * tid + 4 * v);
        zv[4 * v] = t4[0]; zv[4 * v + 1] = t4[1]; zv[4 * v + 2] = t4[2]; zv[4 * v + 3] = t4[3];
      }
      float out[ACT_PPT];
      const int zbase = 2 * (t0 - 4);
#pragma unroll
      for (int r = 0; r < ACT_PPT; ++r) {       // interior form for every output (reads stay inside zs)
#if defined(ACT_ABL) && (ACT_ABL & 4)
        out[r] = zv[2 * r + 8];
        continue;
#endif
        f32x2 a2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 7; ++j)
          a2 = __builtin_elementwise_fma((f32x2){zv[2 * r + 2 + 2 * j], zv[2 * r + 3 + 2 * j]}, fdp[j], a2);
        out[r] = a2[0] + a2[1];
      }
      // outputs whose taps leave [0, 2L-1] exist only in the first and the last tile(s) of a row (uniform test)
      if (t0 == 0 || t0 + ACT_TT + 3 > len) {
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
      if (POUT) {                // natural order through LDS, then 4 consecutive outputs of one phase per thread
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
        const unsigned off = (o0 < ACT_TT && i0 < len) ? (unsigned)(i0 * 4) : 0x80000000u;
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


// ---------------------------------------------------------------------------------------------------
// Wave-private tiles (round 3).  The strip kernel above synchronises its 4 waves three times per tile (x in LDS ->
// barrier -> z in LDS -> barrier -> outputs) and each wave waits for data a third of the time (SQ counters).
// Here a WAVE owns its tile and never waits for another one: lane l holds the 4 consecutive samples x[s0 + 4 l ..]
// (one 16-byte load), takes the 3 + 3 neighbouring samples its filter taps need from lanes l - 1 / l + 1 with DPP
// wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1: register to register, no LDS, no wait), makes its 4 (even, odd)
// pairs of the 2x-rate signal, takes the 3 + 3 neighbouring pairs the same way and writes its 4 outputs as one
// 16-byte store.  Lanes 0-1 and 62-63 only supply halos: a tile = 256 samples read, 240 outputs (AW_OUT) written.
// Arithmetic per sample is the strip kernel's (same taps, same order of the packed FMAs): same bits.
// Phase-major rows (PIN / POUT, the tensors on both sides of a dilated Winograd conv) go through a WAVE-PRIVATE LDS
// transpose instead of the loads / stores above (16-byte chunks of one phase <-> natural order, stride-d scatter /
// gather, conflict-free for odd d): still no block barrier.
// The taps whose index leaves [0, 2 L - 1] (first / last 3 outputs of a row) are replicate-clamped as in the
// reference: such tiles park their pairs in the wave's LDS scratch and redo those outputs tap by tap.
constexpr int AW_OUT = 240;               // outputs per wave tile
constexpr int AW_NT = 4;                  // tiles per wave and block strip (a block = 4 waves walks 16 consecutive tiles)

__device__ __forceinline__ float dpp_from_prev(float v) {      // lane l <- lane l - 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_next(float v) {      // lane l <- lane l + 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

template <bool VEC, bool PIN, bool POUT, bool RAGGED>
__global__ __launch_bounds__(256) void act1d_wave_kernel(const fh_act_group* __restrict__ groups, int batch,
                                                         int channels, int len_arg, int tiles_per_row_arg,
                                                         long long total_tiles, int din, int dout, int n_groups) {
  // per wave: 512 floats of pairs (edge tiles) / natural-order samples (phase-major transposes)
  __shared__ __attribute__((aligned(16))) float scratch[4][528];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const ws = scratch[wave];
  auto lp_of = [](int l, int d) { return ((l + d - 1) / d + 3) & ~3; };

  // position of a flattened tile: (tile in row, channel, batch, group), carried from tile to tile
  struct Pos { int tile, c, bb, gi, tpr, len; };
  auto advance = [&](Pos p, int steps) {
    p.tile += steps;
    while (p.tile >= p.tpr) {
      p.tile -= p.tpr;
      if (++p.c == channels) {
        p.c = 0;
        if (RAGGED) {
          ++p.gi;
          if (p.gi < n_groups) {
            p.len = uni(groups[p.gi].len);
            p.tpr = (p.len + AW_OUT - 1) / AW_OUT;
          }
        } else if (++p.bb == batch) { p.bb = 0; ++p.gi; }
      }
      if (p.gi >= n_groups) break;
    }
    return p;
  };
  const long long g0 = (long long)blockIdx.x * (4 * AW_NT) + wave;      // this wave's first tile; then + 4 per step
  Pos pos;
  if (RAGGED) {
    int cnt = 0;
    for (int base = 0; base < n_groups; base += 64) {
      const int idx = base + lane;
      const bool le = idx < n_groups && (long long)groups[idx].tile_base <= g0;
      cnt += __popcll(__ballot(le));
    }
    pos.gi = uni(cnt - 1);
    pos.bb = 0;
    pos.len = uni(groups[pos.gi].len);
    pos.tpr = (pos.len + AW_OUT - 1) / AW_OUT;
    const unsigned local = (unsigned)g0 - (unsigned)uni(groups[pos.gi].tile_base);
    const unsigned row = local / (unsigned)pos.tpr;
    pos.tile = uni((int)(local - row * (unsigned)pos.tpr));
    pos.c = uni((int)row);
  } else {
    const unsigned g32 = (unsigned)g0;                       // total_tiles < 2^31 (checked by the launcher)
    const unsigned row = g32 / (unsigned)tiles_per_row_arg;
    pos.tile = uni((int)(g32 - row * (unsigned)tiles_per_row_arg));
    const unsigned gb = row / (unsigned)channels;
    pos.c = uni((int)(row - gb * (unsigned)channels));
    pos.gi = uni((int)(gb / (unsigned)batch));
    pos.bb = uni((int)(gb - (unsigned)pos.gi * (unsigned)batch));
    pos.tpr = tiles_per_row_arg;
    pos.len = len_arg;
  }

  // a tile's global side: descriptors of its row, 16-byte request(s) of this lane
  struct Tile {
    __amdgpu_buffer_rsrc_t rx, ry;
    const fh_act_group* G;
    int t0, len, gi;
    float alpha, inv_beta;
  };
  auto tile_of = [&](const Pos& p, bool ok) {
    Tile T;
    T.gi = p.gi;
    T.G = groups + (ok ? p.gi : 0);
    T.len = p.len;
    const int pitch_in = PIN ? din * lp_of(T.len, din) : T.len, pitch_out = POUT ? dout * lp_of(T.len, dout) : T.len;
    const size_t rowi = (size_t)p.bb * channels + p.c;
    T.rx = make_rsrc(uni(T.G->x) + rowi * (size_t)pitch_in, ok ? (unsigned)pitch_in * 4u : 0u);
    T.ry = make_rsrc(uni((const float*)T.G->y) + rowi * (size_t)pitch_out, ok ? (unsigned)pitch_out * 4u : 0u);
    T.t0 = p.tile * AW_OUT;
    T.alpha = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
        make_rsrc(uni(T.G->alpha), ok ? (unsigned)channels * 4u : 0u), (unsigned)p.c * 4u, 0, 0));
    T.inv_beta = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
        make_rsrc(uni(T.G->inv_beta), ok ? (unsigned)channels * 4u : 0u), (unsigned)p.c * 4u, 0, 0));
    return T;
  };
  // plain rows: samples s0 + 4 lane .. + 3 (s0 = t0 - 8, a multiple of 4); outside the row -> 0, patched below.
  // phase-major rows: chunk q = lane + 64 rep = (phase p, 4 consecutive decimated samples), scattered through LDS
  const int nq_in = PIN ? ((256 + din - 1) / din + 3) / 4 : 1;
  const int nq_out = POUT ? ((AW_OUT + dout - 1) / dout + 3) / 4 : 1;
  auto load_tile = [&](const Tile& T, u32x4 (&xr)[2]) {
    const int s0 = T.t0 - 8;
    if (PIN) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int f = lane + 64 * rep;
        const int p = f / nq_in, k = f - p * nq_in;
        const int tb8 = uni(s0 + 8 * din);                         // >= 0
        const int qt = uni(tb8 / din), rt = tb8 - qt * din;
        const int ul = qt - 8 + (rt > p ? 1 : 0);                  // ceil((s0 - p) / din)
        const int u = (ul > 0 ? ul : 0) + 4 * k;
        const int lp_in = lp_of(T.len, din);
        xr[rep] = __builtin_amdgcn_raw_buffer_load_b128(T.rx, p < din ? (unsigned)((p * lp_in + u) * 4) : 0x80000000u, 0, 0);
      }
    } else if (VEC) {
      xr[0] = __builtin_amdgcn_raw_buffer_load_b128(T.rx, (unsigned)((s0 + 4 * lane) * 4), 0, 0);   // (t < 0: out of range)
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        xr[0][e] = __builtin_amdgcn_raw_buffer_load_b32(T.rx, (unsigned)((s0 + 4 * lane + e) * 4), 0, 0);
    }
  };

  bool ok = g0 < total_tiles;
  Tile T = tile_of(pos, ok);
  u32x4 cur[2], nxt[2];
  load_tile(T, cur);
  int gi_prev = -1;
  f32x2 fu2[7], fdp[7];
  float fd[12];
#pragma unroll 1
  for (int it = 0; it < AW_NT; ++it) {
    const bool ok_n = ok && g0 + 4 * (it + 1) < total_tiles && it + 1 < AW_NT;
    if (ok_n) pos = advance(pos, 4);
    const Tile Tn = tile_of(pos, ok_n);
    if (it + 1 < AW_NT) load_tile(Tn, nxt);
    if (ok) {
      const fh_act_group& G = *T.G;
      if (T.gi != gi_prev) {            // taps of the tile's group (wave-uniform; scalar loads)
#pragma unroll
        for (int q = -3; q <= 3; ++q) {
          fu2[q + 3][0] = q <= 2 ? 2.f * G.up_taps[5 - 2 * q] : 0.f;      // (the 2x of UpSample1d folded in: exact)
          fu2[q + 3][1] = q >= -2 ? 2.f * G.up_taps[6 - 2 * q] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) fd[k] = G.down_taps[k];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          fdp[j][0] = j > 0 ? fd[2 * j - 1] : 0.f;
          fdp[j][1] = j < 6 ? fd[2 * j] : 0.f;
        }
        gi_prev = T.gi;
      }
      const int len = T.len, zlast = 2 * len - 1;
      const int s0 = T.t0 - 8;
      const float alpha = T.alpha, inv_beta = T.inv_beta;
      float x[4];
      if (PIN) {          // chunks -> natural order in the wave's LDS (stride din, odd: conflict free) -> own quad
        const int tb8 = uni(s0 + 8 * din);
        const int qt = uni(tb8 / din), rt = tb8 - qt * din;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const int f = lane + 64 * rep;
          const int p = f / nq_in, k = f - p * nq_in;
          const int ul = qt - 8 + (rt > p ? 1 : 0);
          const int u = (ul > 0 ? ul : 0) + 4 * k;
          if (p < din) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int t = (u + e) * din + p;
              if (t >= s0 && t < len && t - s0 < 256) ws[t - s0] = __uint_as_float(cur[rep][e]);
            }
          }
        }
        const f32x4 v = *reinterpret_cast<const f32x4*>(ws + 4 * lane);
        x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = __uint_as_float(cur[0][e]);
      }
      // replicate padding at the row ends (uniform tests; first / last tiles of a row only)
      if (s0 < 0) {                       // t < 0 -> x[0]: element 0 of lane -s0 / 4
        const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[0]), (-s0) >> 2));
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (s0 + 4 * lane + e < 0) x[e] = x0;
      }
      if (s0 + 256 > len) {               // t >= len -> x[len - 1]
        const int le = (len - 1 - s0) >> 2, ee = (len - 1 - s0) & 3;
        const float xsel = ee == 0 ? x[0] : ee == 1 ? x[1] : ee == 2 ? x[2] : x[3];
        const float xe = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xsel), le));
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (s0 + 4 * lane + e >= len) x[e] = xe;
      }
      // X[0..9] = x[i0 - 3 .. i0 + 6], i0 = s0 + 4 lane
      float X[10];
      X[0] = dpp_from_prev(x[1]); X[1] = dpp_from_prev(x[2]); X[2] = dpp_from_prev(x[3]);
      X[3] = x[0]; X[4] = x[1]; X[5] = x[2]; X[6] = x[3];
      X[7] = dpp_from_next(x[0]); X[8] = dpp_from_next(x[1]); X[9] = dpp_from_next(x[2]);
      f32x2 zf[4], arg[4], s2[4], zp[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x2 z = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 7; ++q) z = __builtin_elementwise_fma((f32x2)(X[r + q]), fu2[q], z);
        zf[r] = z;
        arg[r] = z * alpha;
      }
      float amax = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s2[r] = sin_squared2(arg[r]);
        amax = fmaxf(amax, fmaxf(fabsf(arg[r][0]), fabsf(arg[r][1])));     // (NaN falls through, as sinf(NaN))
      }
      if (__builtin_expect(amax >= 32768.f, 0)) {
#pragma unroll 1
        for (int r = 0; r < 4; ++r)
          if (fabsf(arg[r][0]) >= 32768.f || fabsf(arg[r][1]) >= 32768.f) {
            s2[r][0] = sin_squared_slow(arg[r][0]);
            s2[r][1] = sin_squared_slow(arg[r][1]);
          }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) zp[r] = __builtin_elementwise_fma((f32x2)(inv_beta), s2[r], zf[r]);
      // Z[0..9] = pairs of samples i0 - 3 .. i0 + 6
      f32x2 Z[10];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        Z[r] = (f32x2){dpp_from_prev(zp[r + 1][0]), dpp_from_prev(zp[r + 1][1])};
        Z[7 + r] = (f32x2){dpp_from_next(zp[r][0]), dpp_from_next(zp[r][1])};
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) Z[3 + r] = zp[r];
      float out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x2 a2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 7; ++j) a2 = __builtin_elementwise_fma(Z[r + j], fdp[j], a2);
        out[r] = a2[0] + a2[1];
      }
      const int i0 = s0 + 4 * lane;
      const bool edge = T.t0 == 0 || T.t0 + AW_OUT + 3 > len;
      if (edge) {                         // outputs whose taps leave [0, 2 L - 1]: pairs to LDS, redone tap by tap
        *reinterpret_cast<f32x4*>(ws + 8 * lane) = (f32x4){zp[0][0], zp[0][1], zp[1][0], zp[1][1]};
        *reinterpret_cast<f32x4*>(ws + 8 * lane + 4) = (f32x4){zp[2][0], zp[2][1], zp[3][0], zp[3][1]};
        const int zbase = 2 * s0;
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + r;
          if (!(2 * i - 5 >= 0 && 2 * i + 6 <= zlast) && i >= 0 && i < len && lane >= 2 && lane < 62) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
              int m = 2 * i + k - 5;
              m = m < 0 ? 0 : (m > zlast ? zlast : m);
              acc = fmaf(ws[m - zbase], fd[k], acc);
            }
            out[r] = acc;
          }
        }
      }
      const bool mine = lane >= 2 && lane < 62;
      if (POUT) {                // natural order through the wave's LDS, then 4 consecutive outputs of one phase per chunk
        const int lp_out = lp_of(len, dout);
        *reinterpret_cast<f32x4*>(ws + 4 * lane) = (f32x4){out[0], out[1], out[2], out[3]};      // ws[j] = y[s0 + j]
        const int t0 = T.t0;
        const int t_end = (t0 + AW_OUT < len ? t0 + AW_OUT : len) - 1;      // last output of this tile
        const int q0 = uni(t0 / dout), r0 = t0 - q0 * dout;
        const int qe = uni(t_end / dout), re = t_end - qe * dout;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const int f = lane + 64 * rep;
          const int p = f / nq_out, k = f - p * nq_out;
          const int u = q0 + (r0 > p ? 1 : 0) + 4 * k;                      // ceil((t0 - p) / dout) + 4 k
          const int u_hi = p < dout ? qe - (p > re ? 1 : 0) : -1;           // floor((t_end - p) / dout), -1 if t_end < p
          const unsigned off0 = (unsigned)((p * lp_out + u) * 4);
          if (u + 3 <= u_hi) {
            u32x4 ou;
#pragma unroll
            for (int e = 0; e < 4; ++e) ou[e] = __float_as_uint(ws[(u + e) * dout + p - s0]);
            __builtin_amdgcn_raw_buffer_store_b128(ou, T.ry, off0, 0, 0);
          } else {
            for (int e = 0; e < 4 && u + e <= u_hi; ++e)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ws[(u + e) * dout + p - s0]), T.ry, off0 + 4u * e, 0, 0);
          }
        }
      } else if (VEC) {
        const u32x4 ou = {__float_as_uint(out[0]), __float_as_uint(out[1]), __float_as_uint(out[2]), __float_as_uint(out[3])};
        __builtin_amdgcn_raw_buffer_store_b128(ou, T.ry, (mine && i0 < len) ? (unsigned)(i0 * 4) : 0x80000000u, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[r]), T.ry,
                                                (mine && i0 + r < len) ? (unsigned)((i0 + r) * 4) : 0x80000000u, 0, 0);
      }
    }
    T = Tn;
    ok = ok_n;
    cur[0] = nxt[0];
    cur[1] = nxt[1];
  }
}

}  // namespace

extern "C" int fh_sizeof_act_group(void) { return (int)sizeof(fh_act_group); }

extern "C" int fh_act1d_grouped_pm_f32(const fh_act_group* groups, int n_groups, int batch,
                                       int channels, int len, int din, int dout, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && batch > 0 && channels > 0 && len > 0, "fh_act1d_grouped_f32: bad sizes");
  FH_CHECK_ARG(din >= 1 && dout >= 1 && din <= 64 && dout <= 64, "fh_act1d_grouped_pm_f32: bad dilations %d / %d", din, dout);
  const int tiles = fh_cdiv(len, ACT_TT);
  const long long blocks = (long long)n_groups * batch * channels * tiles;
  FH_CHECK_ARG(blocks < (1ll << 31), "fh_act1d_grouped_f32: grid too large");
  // chunks of a tile in phase-major form must fit 2 per thread: ceil(ceil(XS / d) / 4) * d <= 512
  const bool strip_ok = (long long)len * 4 * (din > dout ? din : dout) < (1ll << 31) && din <= 16 && dout <= 16 &&
                        !getenv("FH_ACT_NO_STRIP");
  if (strip_ok && !getenv("FH_ACT_STRIP")) {             // wave-private tiles (act1d_wave_kernel)
    const int wtiles = fh_cdiv(len, AW_OUT);
    const long long total = (long long)n_groups * batch * channels * wtiles;
    FH_CHECK_ARG(total < (1ll << 31), "fh_act1d_grouped_f32: grid too large");
    const unsigned grid = (unsigned)((total + 4 * AW_NT - 1) / (4 * AW_NT));
    const bool vec = (len & 3) == 0;   // rows 16-byte aligned provided the tensors are (checked by the host plan)
#define FH_ACTW_LAUNCH(V, PI, PO)                                                                                    \
  hipLaunchKernelGGL((act1d_wave_kernel<V, PI, PO, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, groups, batch, \
                     channels, len, wtiles, total, din, dout, n_groups)
    if (din > 1 && dout > 1) { if (vec) FH_ACTW_LAUNCH(true, true, true); else FH_ACTW_LAUNCH(false, true, true); }
    else if (din > 1) { if (vec) FH_ACTW_LAUNCH(true, true, false); else FH_ACTW_LAUNCH(false, true, false); }
    else if (dout > 1) { if (vec) FH_ACTW_LAUNCH(true, false, true); else FH_ACTW_LAUNCH(false, false, true); }
    else { if (vec) FH_ACTW_LAUNCH(true, false, false); else FH_ACTW_LAUNCH(false, false, false); }
#undef FH_ACTW_LAUNCH
    FH_CHECK_LAUNCH("fh_act1d_grouped_f32");
    return FH_OK;
  }
  if (strip_ok) {
    const long long strips = (blocks + ACT_NTILE - 1) / ACT_NTILE;
    const bool vec = (len & 3) == 0;   // rows 16-byte aligned provided the tensors are (checked by the host plan)
#define FH_ACT_LAUNCH(V, PI, PO)                                                                              \
  hipLaunchKernelGGL((act1d_strip_kernel<V, PI, PO>), dim3((unsigned)strips), dim3(256), 0, (hipStream_t)stream, \
                     groups, batch, channels, len, tiles, blocks, din, dout, n_groups)
#define FH_ACT_LAUNCH_D(PI, PO, D)                                                                            \
  hipLaunchKernelGGL((act1d_strip_kernel<true, PI, PO, D>), dim3((unsigned)strips), dim3(256), 0, (hipStream_t)stream, \
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
    return FH_OK;
  }
  hipLaunchKernelGGL(act1d_kernel<false>, dim3((unsigned)blocks), dim3(ACT_THREADS), 0, (hipStream_t)stream, groups,
                     batch, channels, len, tiles, din, dout, n_groups);
  FH_CHECK_LAUNCH("fh_act1d_grouped_f32");
  return FH_OK;
}

extern "C" int fh_act_tile_len(void) { return ACT_TT; }

extern "C" int fh_act1d_ragged_f32(const fh_act_group* groups, int n_groups, int channels, int din, int dout,
                                   long long total_tiles, int all_len_mult4, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && channels > 0 && total_tiles > 0 && total_tiles < (1ll << 31),
               "fh_act1d_ragged_f32: bad sizes");
  FH_CHECK_ARG(din >= 1 && dout >= 1 && din <= 64 && dout <= 64, "fh_act1d_ragged_f32: bad dilations %d / %d", din, dout);
  if (din <= 16 && dout <= 16) {      // (row bytes < 2^31: checked by the host plan)
    const long long strips = (total_tiles + ACT_NTILE - 1) / ACT_NTILE;
    const bool vec = all_len_mult4 != 0;
#define FH_ACT_RLAUNCH(V, PI, PO)                                                                                       \
  hipLaunchKernelGGL((act1d_strip_kernel<V, PI, PO, 0, true>), dim3((unsigned)strips), dim3(256), 0, (hipStream_t)stream, \
                     groups, 1, channels, 0, 1, total_tiles, din, dout, n_groups)
    if (din > 1 && dout > 1) FH_ACT_RLAUNCH(false, true, true);
    else if (din > 1) { if (vec) FH_ACT_RLAUNCH(true, true, false); else FH_ACT_RLAUNCH(false, true, false); }
    else if (dout > 1) { if (vec) FH_ACT_RLAUNCH(true, false, true); else FH_ACT_RLAUNCH(false, false, true); }
    else { if (vec) FH_ACT_RLAUNCH(true, false, false); else FH_ACT_RLAUNCH(false, false, false); }
#undef FH_ACT_RLAUNCH
    FH_CHECK_LAUNCH("fh_act1d_ragged_f32");
    return FH_OK;
  }
  hipLaunchKernelGGL(act1d_kernel<true>, dim3((unsigned)total_tiles), dim3(ACT_THREADS), 0, (hipStream_t)stream, groups,
                     1, channels, 0, 1, din, dout, n_groups);
  FH_CHECK_LAUNCH("fh_act1d_ragged_f32");
  return FH_OK;
}

extern "C" int fh_act1d_grouped_f32(const fh_act_group* groups, int n_groups, int batch,
                                    int channels, int len, void* stream) {
  return fh_act1d_grouped_pm_f32(groups, n_groups, batch, channels, len, 1, 1, stream);
}
