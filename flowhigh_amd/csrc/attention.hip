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

// Every query row is reduced as TWO interleaved online-softmax streams -- stream 0 takes the even 32-key tiles,
// stream 1 the odd ones -- merged once at the end by merge_streams().  The two kernel shapes differ only in who
// runs the streams, never in the arithmetic, so a clip gives the same bits whatever else is in the batch:
//   SPLIT = 2 (small grids): 64 queries per block, the two waves of a 32-query tile take one stream each -- twice
//                            the waves and half the dependent MFMA chain per wave when there are fewer query tiles
//                            than SIMDs (B = 1, n = 1000: 512 tiles, 1024 SIMDs);
//   SPLIT = 1 (large grids): 128 queries per block, every wave runs both streams of its tile one after the other
//                            (half the K / V staging traffic per query).
struct Stream {
  f32x16 o0, o1;
  float m, l;
};

__device__ __forceinline__ void merge_streams(Stream& a, const float (&b0)[16], const float (&b1)[16], float mb, float lb) {
  const float m = fmaxf(a.m, mb);                          // (maxima are in the base-2 domain of the tile loop)
  const float c0 = __builtin_amdgcn_exp2f(a.m - m), c1 = __builtin_amdgcn_exp2f(mb - m);      // exp2(-inf) = 0: a stream that saw no key contributes nothing
  a.l = __fmaf_rn(lb, c1, __fmul_rn(a.l, c0));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    a.o0[r] = __fmaf_rn(b0[r], c1, __fmul_rn(a.o0[r], c0));
    a.o1[r] = __fmaf_rn(b1[r], c1, __fmul_rn(a.o1[r], c0));
  }
}

template <int WAVES, int SPLIT>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_kernel(const float* __restrict__ qkv,
                                                               float* __restrict__ out, int n, int heads,
                                                               float scale, const int* __restrict__ seg) {
  constexpr int NT = 64 * WAVES;           // threads
  constexpr int NLD = 1024 / NT;           // float4 of K (and of V) staged per thread and iteration (64 keys)
  constexpr int KT = 32 * KP, VT = 32 * VP;
  __shared__ __attribute__((aligned(16))) float Ks[2 * KT];
  __shared__ __attribute__((aligned(16))) float Vs[2 * VT];
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int inner = heads * 64;
  const size_t ld = (size_t)3 * inner;
  // ragged batch (fh_attention_seg_f32): clip b is rows [seg[2b], seg[2b] + seg[2b+1]) of the token-major tensors;
  // its keys are its own rows only (the reference's key mask, attend.py:127-128, for clips packed without padding)
  size_t row0 = (size_t)b * n;
  if (seg) {
    row0 = (size_t)__builtin_amdgcn_readfirstlane(seg[2 * b]);
    n = __builtin_amdgcn_readfirstlane(seg[2 * b + 1]);
    if ((int)blockIdx.x * (32 * WAVES / SPLIT) >= n) return;        // (block-uniform: before any barrier)
  }
  const float* base = qkv + row0 * ld + h * 64;
  const int qt = wave / SPLIT, sp = wave % SPLIT;      // query tile of the block, stream of this wave (SPLIT = 2)
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

  constexpr int NS = 3 - SPLIT;            // streams this wave runs: 2 (SPLIT = 1) or 1 (SPLIT = 2)
  Stream st[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[i].o0[r] = 0.f; st[i].o1[r] = 0.f; }
    st[i].m = -INFINITY;
    st[i].l = 0.f;
  }

  // one 32-key tile into one stream
  auto tile = [&](Stream& S, int k0, const float* Kt, const float* Vt) {
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
    // online softmax for this lane's query, in base 2 (scores arrive multiplied by scale * log2(e): one v_exp_f32 per
    // probability instead of the libm expf's ~10 instructions -- matrix and vector instructions share the fp32 ALUs, so every
    // one of the ~420 vector instructions per tile cost matrix time: round 6); key of reg r = k0 + (r&3) + 8 (r>>2) + 4 lh
    float mx = -INFINITY;
    if (k0 + 32 <= n) {                              // (whole tile: no key mask -- wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __fmul_rn(s[r], scale);
        mx = fmaxf(mx, s[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = key < n ? __fmul_rn(s[r], scale) : -INFINITY;
        s[r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(S.m, mx);            // finite: every tile has >= 1 valid key
    const float corr = __builtin_amdgcn_exp2f(S.m - m_new);        // exp2(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] - m_new);
      s[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32, 64);
    S.l = __fmaf_rn(S.l, corr, psum);
    S.m = m_new;
    // (the running maximum settles after a few tiles: when no lane's changed, the 32 multiplications by 1 are skipped --
    // x * 1 is exact, so the bits are those of the multiplied form)
    if (__builtin_amdgcn_ballot_w64(corr != 1.f) != 0ull) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { S.o0[r] *= corr; S.o1[r] *= corr; }
    }
    // O^T += V^T P^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v0 = Vt[key * VP + l31];
      const float v1 = Vt[key * VP + 32 + l31];
      S.o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], S.o0, 0, 0, 0);
      S.o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], S.o1, 0, 0, 0);
    }
  };

  // K / V tiles (2 x 512 float4 each) go global -> registers one iteration ahead, registers -> LDS at the top
  // of their own iteration: the loads of the next 64 keys are in flight while these are on the matrix cores
  f32x4 kreg[NLD], vreg[NLD];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + NT * i, key = f >> 4, c4 = f & 15;        // key < 64
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
  for (int kb = 0; kb < n; kb += 64) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + NT * i, key = f >> 4, c4 = f & 15;
      *reinterpret_cast<f32x4*>(Ks + key * KP + 4 * c4) = kreg[i];
      *reinterpret_cast<f32x4*>(Vs + key * VP + 4 * c4) = vreg[i];
    }
    __syncthreads();
    if (kb + 64 < n) load_kv(kb + 64);
    if constexpr (SPLIT == 2) {
      const int k0 = kb + 32 * sp;               // this wave's key tile of the iteration
      if (k0 < n) tile(st[0], k0, Ks + sp * KT, Vs + sp * VT);     // (wave-uniform; the barriers are outside)
    } else {
      tile(st[0], kb, Ks, Vs);
      if (kb + 32 < n) tile(st[1], kb + 32, Ks + KT, Vs + VT);
    }
  }

  if constexpr (SPLIT == 2) {      // stream 1 -> the sp == 0 wave (through the K / V tiles' LDS)
    __syncthreads();
    float* X = Ks + qt * (34 * 64);              // per query tile: 32 O values + m + l per lane
    static_assert((WAVES / SPLIT) * 34 * 64 <= 2 * (KT + VT), "exchange area");
    if (sp == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { X[r * 64 + lane] = st[0].o0[r]; X[(16 + r) * 64 + lane] = st[0].o1[r]; }
      X[32 * 64 + lane] = st[0].m;
      X[33 * 64 + lane] = st[0].l;
    }
    __syncthreads();
    if (sp != 0) return;
    float b0[16], b1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { b0[r] = X[r * 64 + lane]; b1[r] = X[(16 + r) * 64 + lane]; }
    merge_streams(st[0], b0, b1, X[32 * 64 + lane], X[33 * 64 + lane]);
  } else {
    float b0[16], b1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { b0[r] = st[NS - 1].o0[r]; b1[r] = st[NS - 1].o1[r]; }
    merge_streams(st[0], b0, b1, st[NS - 1].m, st[NS - 1].l);
  }
  if (qi < n) {
    const float inv = 1.f / st[0].l;
    float* orow = out + (row0 + qi) * inner + h * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // regs 4g..4g+3 -> d = 8 g + 4 lh + (0..3)
      f32x4 a = {st[0].o0[4 * g] * inv, st[0].o0[4 * g + 1] * inv, st[0].o0[4 * g + 2] * inv, st[0].o0[4 * g + 3] * inv};
      f32x4 c = {st[0].o1[4 * g] * inv, st[0].o1[4 * g + 1] * inv, st[0].o1[4 * g + 2] * inv, st[0].o1[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(orow + 8 * g + 4 * lh) = a;
      *reinterpret_cast<f32x4*>(orow + 32 + 8 * g + 4 * lh) = c;
    }
  }
}

}  // namespace

namespace {
int launch_attention(const float* qkv, float* out, const int* seg, int batch, int n, int heads, float scale, void* stream) {
  scale *= 1.44269504088896340736f;        // the kernel's softmax runs in base 2: exp(x) = exp2(x log2(e))
  if ((long long)fh_cdiv(n, 128) * heads * batch >= 512) {
    dim3 grid(fh_cdiv(n, 128), heads, batch);
    hipLaunchKernelGGL((attention_kernel<4, 1>), grid, dim3(256), 0, (hipStream_t)stream, qkv, out, n, heads, scale, seg);
  } else {
    dim3 grid(fh_cdiv(n, 64), heads, batch);
    hipLaunchKernelGGL((attention_kernel<4, 2>), grid, dim3(256), 0, (hipStream_t)stream, qkv, out, n, heads, scale, seg);
  }
  return 0;
}
}  // namespace

extern "C" int fh_attention_f32(const float* qkv, float* out, int batch, int n, int heads,
                                float scale, void* stream) {
  FH_CHECK_ARG(qkv && out && batch > 0 && n > 0 && heads > 0, "fh_attention_f32: bad args");
  launch_attention(qkv, out, nullptr, batch, n, heads, scale, stream);
  FH_CHECK_LAUNCH("fh_attention_f32");
  return FH_OK;
}

extern "C" int fh_attention_seg_f32(const float* qkv, float* out, const int* seg, int n_seg, int max_n, int heads,
                                    float scale, void* stream) {
  FH_CHECK_ARG(qkv && out && seg && n_seg > 0 && max_n > 0 && heads > 0, "fh_attention_seg_f32: bad args");
  launch_attention(qkv, out, seg, n_seg, max_n, heads, scale, stream);
  FH_CHECK_LAUNCH("fh_attention_seg_f32");
  return FH_OK;
}
