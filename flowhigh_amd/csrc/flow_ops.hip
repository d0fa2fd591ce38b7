// HBM-bound pointwise / reduction kernels of the FLowHigh transformer.
//   fh_dwconv_gelu_res_f32 : ConvPositionEmbed + residual   models/transformer.py:16-46, flow.py:240
//   fh_rmsnorm_f32         : AdaptiveRMSNorm / RMSNorm      models/transformer.py:49-88
//   fh_qknorm_rope_f32     : MultiheadRMSNorm + RoPE        models/attend.py:144-151,179-184, pos_emb.py:53-59
// All tensors are token-major [B*n, dim]; consecutive lanes touch consecutive features
// (16-byte vector accesses), row reductions are 64-lane butterflies, no LDS.
#include "fh_common.h"

namespace {

// Block = 128 channels x 32 tokens.  The x slab [32 + ksz - 1][128] is staged in LDS once (every
// element is used by ksz outputs); thread = (channel quad, token lane) makes 4 tokens x 4 channels
// with 16-byte LDS reads; weights are read as [ksz][dim] (transposed on the host) so that a tap is
// one coalesced 16-byte load per thread.
constexpr int DW_TOK = 32;
constexpr int DW_CH = 128;
constexpr int DW_KMAX = 63;

__global__ __launch_bounds__(256) void dwconv_gelu_res_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ wt,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ y, int n, int dim,
                                                              int ksz, const int* __restrict__ seg) {
  extern __shared__ __attribute__((aligned(16))) float xs[];     // [DW_TOK + ksz - 1][DW_CH]
  const int c0 = blockIdx.x * DW_CH;
  const int tok0 = blockIdx.y * DW_TOK;
  const int b = blockIdx.z;
  const int half = ksz / 2;
  const int rows = DW_TOK + ksz - 1;
  size_t row0 = (size_t)b * n;
  if (seg) {             // ragged batch: clip b = rows [seg[2b], + seg[2b+1]); zero padding at ITS ends (transformer.py:35-44)
    row0 = (size_t)seg[2 * b];
    n = seg[2 * b + 1];
    if (tok0 >= n) return;
  }
  const float* xb = x + row0 * dim;
  float* yb = y + row0 * dim;
  const int tid = threadIdx.x;
  // stage: 32 threads per row (128 channels = 32 float4), 8 rows per pass; zero outside [0, n)
  for (int r = tid >> 5; r < rows; r += 8) {
    const int t = tok0 + r - half;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t >= 0 && t < n) v = *reinterpret_cast<const f32x4*>(xb + (size_t)t * dim + c0 + 4 * (tid & 31));
    *reinterpret_cast<f32x4*>(xs + r * DW_CH + 4 * (tid & 31)) = v;
  }
  __syncthreads();
  const int cq = tid & 31, tl = tid >> 5;            // channel quad, token lane: tokens tl, tl + 8, ...
  const int c4 = c0 + 4 * cq;
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c4);
  f32x4 acc[4] = {bv, bv, bv, bv};
  for (int j = 0; j < ksz; ++j) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wt + (size_t)j * dim + c4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + (tl + 8 * q + j) * DW_CH + 4 * cq);
      acc[q][0] = fmaf(w4[0], xv[0], acc[q][0]);
      acc[q][1] = fmaf(w4[1], xv[1], acc[q][1]);
      acc[q][2] = fmaf(w4[2], xv[2], acc[q][2]);
      acc[q][3] = fmaf(w4[3], xv[3], acc[q][3]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = tok0 + tl + 8 * q;
    if (t >= n) continue;
    const f32x4 xc = *reinterpret_cast<const f32x4*>(xs + (tl + 8 * q + half) * DW_CH + 4 * cq);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = xc[e] + gelu_erf(acc[q][e]);
    *reinterpret_cast<f32x4*>(yb + (size_t)t * dim + c4) = o;
  }
}

// one wave per row; dim % 256 == 0, dim <= 4096
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      float* __restrict__ y, int rows, int dim,
                                                      float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * dim;
  float* yr = y + (size_t)row * dim;
  f32x4 v[16];
  const int nv = dim >> 8;   // float4 per lane
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < nv) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4);
      ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
  }
  ss = wave_sum(ss);
  const float inv = scale / fmaxf(sqrtf(ss), 1e-12f);     // F.normalize eps
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < nv) {
      const int c = (i * 64 + lane) * 4;
      f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      f32x4 o;
      if (beta) {
        f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[i][e] * inv * g[e] + bt[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[i][e] * inv * g[e];
      }
      *reinterpret_cast<f32x4*>(yr + c) = o;
    }
  }
}

// one wave per (token, head, q|k); lane = feature d of the 64-wide head
__global__ __launch_bounds__(256) void qknorm_rope_kernel(float* __restrict__ qkv,
                                                          const float* __restrict__ gq,
                                                          const float* __restrict__ gk,
                                                          const float* __restrict__ cos_t,
                                                          const float* __restrict__ sin_t, int n,
                                                          int heads, long long total, const int* __restrict__ seg) {
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (item >= total) return;
  const int which = (int)(item % 2);                 // 0 = q, 1 = k
  const int h = (int)((item / 2) % heads);
  long long row = item / (2 * heads);                // b * n + pos
  const int pos = (int)(row % n);
  if (seg) {             // ragged batch: position `pos` of clip b, if it has one (n = longest clip)
    const int b = (int)(row / n);
    if (pos >= seg[2 * b + 1]) return;
    row = (long long)seg[2 * b] + pos;
  }
  const int inner = heads * 64;
  float* p = qkv + (size_t)row * (3 * inner) + which * inner + h * 64;
  const float v = p[lane];
  const float ss = wave_sum(v * v);
  const float g = (which ? gk : gq)[h * 64 + lane];
  // F.normalize(x) * gamma * scale  (attend.py:150): (x / max(||x||, eps)) * gamma * 8
  const float t = v / fmaxf(sqrtf(ss), 1e-12f) * g * 8.0f;
  const float other = __shfl_xor(t, 32, 64);         // rotate_half partner
  const float rot = lane < 32 ? -other : other;      // cat(-x2, x1)
  const float cs = cos_t[pos * 32 + (lane & 31)];
  const float sn = sin_t[pos * 32 + (lane & 31)];
  p[lane] = t * cs + rot * sn;
}

}  // namespace

extern "C" int fh_dwconv_gelu_res_f32(const float* x, const float* w, const float* bias, float* y,
                                      int batch, int n, int dim, int ksz, void* stream) {
  FH_CHECK_ARG(x && w && bias && y && batch > 0 && n > 0, "fh_dwconv_gelu_res_f32: bad args");
  FH_CHECK_ARG(dim % DW_CH == 0 && (ksz & 1) && ksz <= DW_KMAX, "fh_dwconv_gelu_res_f32: dim %d / ksz %d unsupported", dim, ksz);
  dim3 grid(dim / DW_CH, fh_cdiv(n, DW_TOK), batch);
  const size_t lds = (size_t)(DW_TOK + ksz - 1) * DW_CH * sizeof(float);
  hipLaunchKernelGGL(dwconv_gelu_res_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, y, n,
                     dim, ksz, (const int*)nullptr);
  FH_CHECK_LAUNCH("fh_dwconv_gelu_res_f32");
  return FH_OK;
}

extern "C" int fh_dwconv_gelu_res_seg_f32(const float* x, const float* w, const float* bias, float* y,
                                          const int* seg, int n_seg, int max_n, int dim, int ksz, void* stream) {
  FH_CHECK_ARG(x && w && bias && y && seg && n_seg > 0 && max_n > 0, "fh_dwconv_gelu_res_seg_f32: bad args");
  FH_CHECK_ARG(dim % DW_CH == 0 && (ksz & 1) && ksz <= DW_KMAX, "fh_dwconv_gelu_res_seg_f32: dim %d / ksz %d unsupported", dim, ksz);
  dim3 grid(dim / DW_CH, fh_cdiv(max_n, DW_TOK), n_seg);
  const size_t lds = (size_t)(DW_TOK + ksz - 1) * DW_CH * sizeof(float);
  hipLaunchKernelGGL(dwconv_gelu_res_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, y, max_n,
                     dim, ksz, seg);
  FH_CHECK_LAUNCH("fh_dwconv_gelu_res_seg_f32");
  return FH_OK;
}

extern "C" int fh_rmsnorm_f32(const float* x, const float* gamma, const float* beta, float* y,
                              int rows, int dim, void* stream) {
  FH_CHECK_ARG(x && gamma && y && rows > 0, "fh_rmsnorm_f32: bad args");
  FH_CHECK_ARG(dim % 256 == 0 && dim <= 4096, "fh_rmsnorm_f32: dim %d unsupported", dim);
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(fh_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                     beta, y, rows, dim, sqrtf((float)dim));
  FH_CHECK_LAUNCH("fh_rmsnorm_f32");
  return FH_OK;
}

extern "C" int fh_qknorm_rope_f32(float* qkv, const float* gq, const float* gk, const float* cos_t,
                                  const float* sin_t, int batch, int n, int heads, void* stream) {
  FH_CHECK_ARG(qkv && gq && gk && cos_t && sin_t && batch > 0 && n > 0 && heads > 0, "fh_qknorm_rope_f32: bad args");
  const long long total = (long long)batch * n * heads * 2;
  hipLaunchKernelGGL(qknorm_rope_kernel, dim3(fh_cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, qkv,
                     gq, gk, cos_t, sin_t, n, heads, total, (const int*)nullptr);
  FH_CHECK_LAUNCH("fh_qknorm_rope_f32");
  return FH_OK;
}

extern "C" int fh_qknorm_rope_seg_f32(float* qkv, const float* gq, const float* gk, const float* cos_t,
                                      const float* sin_t, const int* seg, int n_seg, int max_n, int heads,
                                      void* stream) {
  FH_CHECK_ARG(qkv && gq && gk && cos_t && sin_t && seg && n_seg > 0 && max_n > 0 && heads > 0,
               "fh_qknorm_rope_seg_f32: bad args");
  const long long total = (long long)n_seg * max_n * heads * 2;
  hipLaunchKernelGGL(qknorm_rope_kernel, dim3(fh_cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, qkv,
                     gq, gk, cos_t, sin_t, max_n, heads, total, seg);
  FH_CHECK_LAUNCH("fh_qknorm_rope_seg_f32");
  return FH_OK;
}
