// Streaming-softmax attention in fp32 on the gfx950 matrix cores (dim_head = 64).
//
// Replaces Attend.forward, /root/reference/src/flowhigh/models/attend.py:102-139
//   sim = einsum(q, k) * scale ; attn = softmax(sim) ; out = einsum(attn, v)
// which materialises [B, 16, n, n] three times (64 MB per clip at n = 1000); here the scores
// live only in MFMA accumulators.
//
// Per wave: 32 queries; per block (4 waves): 128 queries of one (batch, head); key/value tiles of
// 32 keys are staged in LDS and shared by the 4 waves.
//   S^T = K Q^T  (M = key, N = query, K = d):  A = K tile from LDS (ds_read_b128, 4 k-steps each),
//                 B = Q fragments held in 32 VGPRs for the whole kernel.
//   The accumulator then holds, for the lane's query (col = lane & 31), 16 keys per lane half:
//   the row softmax is 16 in-register ops + one exchange with lane ^ 32, and the probabilities are
//   already the B operand of the next product (no LDS round trip, no conversion):
//   O^T = V^T P^T (M = d, N = query, K = key): k-step r pairs keys (r&3)+8(r>>2) and that + 4,
//                 A = V^T read from the LDS V tile with the same key order.
#include <math.h>

#include "fh_common.h"

namespace {

constexpr int KP = 68;   // K tile pitch (floats): b128 reads conflict free
constexpr int VP = 64;

// WAVES = 4, SPLIT = 1: 128 queries per block.  SPLIT = 2 (small grids): 64 queries per block, the two waves of a
// 32-query tile take alternate 32-key tiles and merge their (max, sum, O) at the end -- twice the waves and half
// the dependent MFMA chain per wave when there are fewer query tiles than SIMDs (B = 1: 512 tiles, 1024 SIMDs).
template <int WAVES, int SPLIT>
__global__ __launch_bounds__(64 * WAVES) void attention_kernel(const float* __restrict__ qkv,
                                                               float* __restrict__ out, int n, int heads,
                                                               float scale) {
  constexpr int NT = 64 * WAVES;           // threads
  constexpr int NLD = 512 * SPLIT / NT;    // float4 of K (and of V) staged per thread and iteration
  constexpr int KT = 32 * KP, VT = 32 * VP;
  __shared__ __attribute__((aligned(16))) float Ks[SPLIT * KT];
  __shared__ __attribute__((aligned(16))) float Vs[SPLIT * VT];
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int inner = heads * 64;
  const size_t ld = (size_t)3 * inner;
  const float* base = qkv + (size_t)b * n * ld + h * 64;
  const int qt = wave / SPLIT, sp = wave % SPLIT;      // query tile of the block, key-split index
  const int q0 = blockIdx.x * (32 * WAVES / SPLIT) + qt * 32;
  const int qi = q0 + l31;

  // Q fragments: qf[q'][e] = Q[qi][8 q' + 4 lh + e]
  f32x4 qf[8];
#pragma unroll
  for (int qq = 0; qq < 8; ++qq) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (qi < n) v = *reinterpret_cast<const f32x4*>(base + (size_t)qi * ld + 4 * (2 * qq + lh));
    qf[qq] = v;
  }

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  // K / V tiles (512 float4 each) go global -> registers one tile ahead, registers -> LDS at the top of
  // their own iteration: the loads of tile i + 1 are in flight while tile i is on the matrix cores
  f32x4 kreg[NLD], vreg[NLD];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + NT * i, key = f >> 4, c4 = f & 15;        // key < 32 SPLIT
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (k0 + key < n) {
        const float* rowp = base + (size_t)(k0 + key) * ld + 4 * c4;
        kv = *reinterpret_cast<const f32x4*>(rowp + inner);
        vv = *reinterpret_cast<const f32x4*>(rowp + 2 * inner);
      }
      kreg[i] = kv;
      vreg[i] = vv;
    }
  };
  load_kv(0);
  for (int kb = 0; kb < n; kb += 32 * SPLIT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + NT * i, key = f >> 4, c4 = f & 15;
      *reinterpret_cast<f32x4*>(Ks + key * KP + 4 * c4) = kreg[i];
      *reinterpret_cast<f32x4*>(Vs + key * VP + 4 * c4) = vreg[i];
    }
    __syncthreads();
    if (kb + 32 * SPLIT < n) load_kv(kb + 32 * SPLIT);
    const int k0 = kb + 32 * sp;                 // this wave's key tile of the iteration
    const float* Kt = Ks + sp * KT;
    const float* Vt = Vs + sp * VT;
    if (k0 >= n) continue;                       // (wave-uniform; the barriers above are outside)

    // S^T tile
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      f32x4 kf = *reinterpret_cast<const f32x4*>(Kt + l31 * KP + 4 * (2 * qq + lh));
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[qq][e], s, 0, 0, 0);
    }

    // online softmax for this lane's query; key of reg r = k0 + (r&3) + 8 (r>>2) + 4 lh
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = s[r] * scale;
      v = key < n ? v : -INFINITY;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);          // finite: every tile has >= 1 valid key
    const float corr = expf(m_run - m_new);        // exp(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = expf(s[r] - m_new);
      s[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }

    // O^T += V^T P^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v0 = Vt[key * VP + l31];
      const float v1 = Vt[key * VP + 32 + l31];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o1, 0, 0, 0);
    }
  }

  if constexpr (SPLIT > 1) {       // merge the key-split partial results into the sp == 0 wave (through the K / V tiles' LDS)
    __syncthreads();
    float* X = Ks + qt * (34 * 64);              // per query tile: 32 O values + m + l per lane (SPLIT == 2)
    static_assert(SPLIT <= 2 && (WAVES / SPLIT) * 34 * 64 <= SPLIT * (KT + VT), "exchange area");
    if (sp == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { X[r * 64 + lane] = o0[r]; X[(16 + r) * 64 + lane] = o1[r]; }
      X[32 * 64 + lane] = m_run;
      X[33 * 64 + lane] = l_run;
    }
    __syncthreads();
    if (sp != 0) return;
    const float m1 = X[32 * 64 + lane], l1 = X[33 * 64 + lane];
    const float m = fmaxf(m_run, m1);
    const float c0 = expf(m_run - m), c1 = expf(m1 - m);      // exp(-inf) = 0: a wave that saw no key contributes nothing
    l_run = l_run * c0 + l1 * c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0[r] = o0[r] * c0 + X[r * 64 + lane] * c1;
      o1[r] = o1[r] * c0 + X[(16 + r) * 64 + lane] * c1;
    }
  }
  if (qi < n) {
    const float inv = 1.f / l_run;
    float* orow = out + ((size_t)b * n + qi) * inner + h * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // regs 4g..4g+3 -> d = 8 g + 4 lh + (0..3)
      f32x4 a = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
      f32x4 c = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(orow + 8 * g + 4 * lh) = a;
      *reinterpret_cast<f32x4*>(orow + 32 + 8 * g + 4 * lh) = c;
    }
  }
}

}  // namespace

extern "C" int fh_attention_f32(const float* qkv, float* out, int batch, int n, int heads,
                                float scale, void* stream) {
  FH_CHECK_ARG(qkv && out && batch > 0 && n > 0 && heads > 0, "fh_attention_f32: bad args");
  if ((long long)fh_cdiv(n, 128) * heads * batch >= 512) {
    dim3 grid(fh_cdiv(n, 128), heads, batch);
    hipLaunchKernelGGL((attention_kernel<4, 1>), grid, dim3(256), 0, (hipStream_t)stream, qkv, out, n, heads, scale);
  } else {
    dim3 grid(fh_cdiv(n, 64), heads, batch);
    hipLaunchKernelGGL((attention_kernel<4, 2>), grid, dim3(256), 0, (hipStream_t)stream, qkv, out, n, heads, scale);
  }
  FH_CHECK_LAUNCH("fh_attention_f32");
  return FH_OK;
}
