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
// Block = 256 threads, one (group, batch, channel) row segment of TT = 1018 outputs:
//   phase 1: x[t0-6 .. t0+TT+5] -> LDS                       (TT + 12 floats, clamped indices)
//   phase 2: each thread makes 4 (even, odd) pairs of z -> LDS   (TT + 6 pairs = i in [t0-3, t0+TT+2])
//   phase 3: each thread makes <= 4 outputs from 7 ds_read_b64 each.
#include "fh_common.h"

namespace {

// sin^2(a) without libm's sinf (the activation is VALU-bound on it: two calls per sample at the 2x
// rate).  Cody-Waite reduction by pi/2 in three fused steps (exact products for |k| < 2^15), then the
// odd Taylor polynomial of sin to r^9 on |r| <= pi/4 (truncation 2e-9) and sin^2 = s^2 for even k,
// 1 - s^2 for odd k.  Absolute error <= ~2e-7, the same size as squaring a 1-ulp sinf.
__device__ __forceinline__ float sin_squared(float a) {
  if (fabsf(a) >= 32768.f) {            // never reached by sane activations; keeps the result exact-ish
    const float s = sinf(a);
    return s * s;
  }
  const float k = rintf(a * 0.63661977236758134308f);
  float r = fmaf(k, -1.5703125f, a);
  r = fmaf(k, -4.837512969970703125e-4f, r);
  r = fmaf(k, -7.549789948768648e-8f, r);
  const float r2 = r * r;
  float p = fmaf(r2, 2.7557314297e-06f, -1.9841270114e-04f);
  p = fmaf(r2, p, 8.3333337680e-03f);
  p = fmaf(r2, p, -1.6666667163e-01f);
  const float s = fmaf(r * r2, p, r);
  const float s2 = s * s;
  return (static_cast<int>(k) & 1) ? 1.0f - s2 : s2;
}

// The kernel is latency bound on its global loads (each block reads its window once, up front), so
// a block covers 4 pairs per thread: all of a thread's loads are issued back to back and ~4 KB per
// block (x 8 resident blocks per CU) are in flight.
constexpr int ACT_PPT = 4;                   // z pairs per thread
constexpr int ACT_PAIRS = 256 * ACT_PPT;     // 1024 z pairs
constexpr int ACT_TT = ACT_PAIRS - 6;        // 1018 outputs per block
constexpr int ACT_XW = ACT_TT + 12;          // 1030 staged inputs
constexpr int ACT_XLD = (ACT_XW + 255) / 256;

__global__ __launch_bounds__(256) void act1d_kernel(const fh_act_group* __restrict__ groups,
                                                    int batch, int channels, int len,
                                                    int tiles_per_row) {
  __shared__ float xs[ACT_XW + 2];
  __shared__ __attribute__((aligned(8))) float zs[2 * ACT_PAIRS];

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

  // phase 1: clamped (replicate) input window; loads first, LDS writes after
  float xin[ACT_XLD];
#pragma unroll
  for (int i = 0; i < ACT_XLD; ++i) {
    int t = t0 - 6 + tid + 256 * i;
    t = t < 0 ? 0 : (t > len - 1 ? len - 1 : t);
    xin[i] = x[t];
  }
#pragma unroll
  for (int i = 0; i < ACT_XLD; ++i) {
    const int j = tid + 256 * i;
    if (j < ACT_XW) xs[j] = xin[i];
  }
  __syncthreads();

  // phase 2: z pairs for i = t0 - 3 + p, p in [0, 512)
  float fu[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) fu[k] = G.up_taps[k];
  const int zlast = 2 * len - 1;
#pragma unroll
  for (int rep = 0; rep < ACT_PPT; ++rep) {
    const int p = tid + 256 * rep;          // pair index; sample i = t0 - 3 + p
    // x[i+q] is xs[p + q + 3]  (xs[j] <-> t0 - 6 + j)
    float xv[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) xv[q] = xs[p + q];      // x[i-3 .. i+3]
    float ze = 0.f, zo = 0.f;
#pragma unroll
    for (int q = -3; q <= 2; ++q) ze = fmaf(xv[q + 3], fu[5 - 2 * q], ze);
#pragma unroll
    for (int q = -2; q <= 3; ++q) zo = fmaf(xv[q + 3], fu[6 - 2 * q], zo);
    ze *= 2.f;
    zo *= 2.f;
    ze = ze + inv_beta * sin_squared(ze * alpha);
    zo = zo + inv_beta * sin_squared(zo * alpha);
    // positions outside [0, 2L-1] are never used directly: phase 3 clamps its index instead
    zs[2 * p] = ze;
    zs[2 * p + 1] = zo;
  }
  __syncthreads();

  // phase 3: y[i] = sum_k z[clamp(2i + k - 5)] f_dn[k];  z[m] is zs[m - 2 (t0 - 3)]
  float fd[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) fd[k] = G.down_taps[k];
  const int zbase = 2 * (t0 - 3);
#pragma unroll
  for (int rep = 0; rep < ACT_PPT; ++rep) {
    const int o = tid + 256 * rep;
    const int i = t0 + o;
    if (o >= ACT_TT || i >= len) continue;
    float acc = 0.f;
    if (2 * i - 5 >= 0 && 2 * i + 6 <= zlast) {
      // interior: 14 contiguous values zs[2o .. 2o+13], taps use [1 .. 12]
      const float2* zp = reinterpret_cast<const float2*>(zs + 2 * o);
      float v[14];
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        float2 t2 = zp[q];
        v[2 * q] = t2.x;
        v[2 * q + 1] = t2.y;
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) acc = fmaf(v[k + 1], fd[k], acc);
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        int m = 2 * i + k - 5;
        m = m < 0 ? 0 : (m > zlast ? zlast : m);
        acc = fmaf(zs[m - zbase], fd[k], acc);
      }
    }
    y[i] = acc;
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
