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

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// sin^2(a) without libm's sinf, two values at a time.  Cody-Waite reduction by pi/2 in three fused
// steps (exact products for |k| < 2^15), then the odd Taylor polynomial of sin to r^9 on
// |r| <= pi/4 (truncation 2e-9); sin^2 = s^2 for even k, 1 - s^2 for odd k.  Absolute error
// <= 1.2e-7 (checked on the host against float64 over |a| < 3e4), the same size as squaring a
// 1-ulp sinf.  |a| >= 32768 is patched by the caller.
__device__ __forceinline__ f32x2 sin_squared2(f32x2 a) {
  const f32x2 t = a * 0.63661977236758134308f;
  const f32x2 k = {rintf(t[0]), rintf(t[1])};
  f32x2 r = __builtin_elementwise_fma(k, (f32x2)(-1.5703125f), a);
  r = __builtin_elementwise_fma(k, (f32x2)(-4.837512969970703125e-4f), r);
  r = __builtin_elementwise_fma(k, (f32x2)(-7.549789948768648e-8f), r);
  const f32x2 r2 = r * r;
  f32x2 p = __builtin_elementwise_fma(r2, (f32x2)(2.7557314297e-06f), (f32x2)(-1.9841270114e-04f));
  p = __builtin_elementwise_fma(r2, p, (f32x2)(8.3333337680e-03f));
  p = __builtin_elementwise_fma(r2, p, (f32x2)(-1.6666667163e-01f));
  const f32x2 s = __builtin_elementwise_fma(r * r2, p, r);
  const f32x2 s2 = s * s;
  f32x2 o;
  o[0] = (static_cast<int>(k[0]) & 1) ? 1.0f - s2[0] : s2[0];
  o[1] = (static_cast<int>(k[1]) & 1) ? 1.0f - s2[1] : s2[1];
  return o;
}

__device__ __noinline__ float sin_squared_slow(float a) {   // huge arguments only (never in practice)
  const float s = sinf(a);
  return s * s;
}

constexpr int ACT_PPT = 4;                   // z pairs (and outputs) per thread, consecutive
constexpr int ACT_PAIRS = 256 * ACT_PPT;     // z pairs of a tile: samples i = t0 - 4 + p
constexpr int ACT_TT = ACT_PAIRS - 8;        // outputs per tile (multiple of 8: tiles start 16-byte aligned)
constexpr int ACT_XS = ACT_TT + 16;          // staged inputs x[t0-8 .. t0+TT+7]
constexpr int ACT_XF4 = ACT_XS / 4;          // ... as float4s

// One block = one (group, batch, channel, tile).  Thread t owns pairs 4t .. 4t+3 and outputs
// 4t .. 4t+3: every LDS access is a 16-byte vector, and when rows are 16-byte aligned
// (len % 4 == 0) so is every global access -- 4-byte-per-lane loads ran this kernel at 2.5 TB/s.
__global__ __launch_bounds__(256) void act1d_kernel(const fh_act_group* __restrict__ groups,
                                                    int batch, int channels, int len,
                                                    int tiles_per_row) {
  __shared__ __attribute__((aligned(16))) float xs[ACT_PAIRS + 16];
  __shared__ __attribute__((aligned(16))) float zs[2 * ACT_PAIRS + 16];

  const int tile = blockIdx.x % tiles_per_row;
  const int row = blockIdx.x / tiles_per_row;          // (g * batch + b) * channels + c
  const int c = row % channels;
  const int gb = row / channels;
  const fh_act_group& G = groups[gb / batch];
  const int b = gb % batch;
  const size_t base = ((size_t)b * channels + c) * len;
  const float* __restrict__ x = G.x + base;
  float* __restrict__ y = G.y + base;
  const float alpha = G.alpha[c];
  const float inv_beta = G.inv_beta[c];
  const int t0 = tile * ACT_TT;
  const int tid = threadIdx.x;
  const int zlast = 2 * len - 1;
  const bool vec = (len & 3) == 0 && ((((size_t)G.x) | ((size_t)G.y)) & 15) == 0;

  // phase 1: xs[j] = x[clamp(t0 - 8 + j)], j < XS; thread t stages float4 #t, #256+t, ...
#pragma unroll
  for (int rep = 0; rep < (ACT_XF4 + 255) / 256; ++rep) {
    const int f = tid + 256 * rep;
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
    fu2[q + 3][0] = q <= 2 ? G.up_taps[5 - 2 * q] : 0.f;
    fu2[q + 3][1] = q >= -2 ? G.up_taps[6 - 2 * q] : 0.f;
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
      z = z * 2.f;
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
  const int o0 = ACT_PPT * tid;
  const int i0 = t0 + o0;
  if (o0 >= ACT_TT || i0 >= len) return;
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
#pragma unroll
      for (int k = 0; k < 12; ++k) acc = fmaf(zv[2 * r + 3 + k], fd[k], acc);
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

}  // namespace

extern "C" int fh_sizeof_act_group(void) { return (int)sizeof(fh_act_group); }

extern "C" int fh_act1d_grouped_f32(const fh_act_group* groups, int n_groups, int batch,
                                    int channels, int len, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && batch > 0 && channels > 0 && len > 0, "fh_act1d_grouped_f32: bad sizes");
  const int tiles = fh_cdiv(len, ACT_TT);
  const long long blocks = (long long)n_groups * batch * channels * tiles;
  FH_CHECK_ARG(blocks < (1ll << 31), "fh_act1d_grouped_f32: grid too large");
  hipLaunchKernelGGL(act1d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, groups,
                     batch, channels, len, tiles);
  FH_CHECK_LAUNCH("fh_act1d_grouped_f32");
  return FH_OK;
}
