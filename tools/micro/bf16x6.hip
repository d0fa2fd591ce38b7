// Micro-benchmark: what would an fp32-grade contraction on the BF16 matrix cores cost?
//
// fp32 operands split exactly into three bf16 pieces (x = h + m + l, 8 significant bits each), product summed over
// the 6 leading piece pairs (hh, hm, mh, mm, hl, lh): dropped terms <= ~2^-24 |a||b| per product, fp32 accumulation
// in the MFMA.  One fp32 k-block of 16 = 6 x v_mfma_f32_32x32x16_bf16 (32 cycles each) against 8 x
// v_mfma_f32_32x32x2_f32 (64 cycles each): 0.375 of the matrix-pipe cycles.  The price: the B operand (activations)
// has to be split on the VALU: per 2 values cvt_pk, shl, and, pk_sub twice + a last cvt_pk = 9 instructions.
//
// Both loops below mimic the operand flow of conv_wino_kernel's K loop -- B values read from LDS (fp32), a 4-term
// "transform" in packed fp32 math, A values streamed from global memory (L2 resident), 12 waves per block, one block
// per CU -- so that the comparison includes the VALU / LDS work beside the MFMAs, not just the MFMA rate.
//   ./bf16x6 [iters]      prints fp32-equivalent TFLOP/s of both forms for a few wave tiles, and the error of the split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 768;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {          // v_cvt_pk_bf16_f32 (round to nearest even)
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// 8 fp32 values -> three bf16x8 pieces (exact split: v = h + m + l up to 2^-27 |v|)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  unsigned hp[4], mp[4], lp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    hp[i] = pack_bf16(a, b);
    const float ra = a - __uint_as_float(hp[i] << 16), rb = b - __uint_as_float(hp[i] & 0xffff0000u);
    mp[i] = pack_bf16(ra, rb);
    const float sa = ra - __uint_as_float(mp[i] << 16), sb = rb - __uint_as_float(mp[i] & 0xffff0000u);
    lp[i] = pack_bf16(sa, sb);
  }
  h = __builtin_bit_cast(bf16x8, (u32x4){hp[0], hp[1], hp[2], hp[3]});
  m = __builtin_bit_cast(bf16x8, (u32x4){mp[0], mp[1], mp[2], mp[3]});
  l = __builtin_bit_cast(bf16x8, (u32x4){lp[0], lp[1], lp[2], lp[3]});
}

// ---- fp32 form: per k-block of 16: NT x (8 LDS float2 reads, 16 packed transform ops), 8 MT NT MFMAs --------------
template <int MT, int NT>
__global__ __launch_bounds__(THREADS) void loop_f32(const float* __restrict__ A, float* __restrict__ out, int kblocks) {
  __shared__ __attribute__((aligned(16))) float slab[16 * 4 * 72 * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * 4 * 72 * 2; i += THREADS) slab[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 113) - 0.05f;
  __syncthreads();
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float* ap = A + ((size_t)(blockIdx.x % 8) * 12 + wave) * 4096 + lane * 8;
  const f32x2 c0 = {4.f, 4.f}, c1 = {-5.f, -5.f}, c2 = {1.f, 1.f}, c3 = {0.5f, 0.5f};
  for (int kb = 0; kb < kblocks; ++kb) {
    f32x4 a[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      a[mt][0] = *reinterpret_cast<const f32x4*>(ap + ((kb & 7) * MT + mt) * 512);
      a[mt][1] = *reinterpret_cast<const f32x4*>(ap + ((kb & 7) * MT + mt) * 512 + 4);
    }
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {              // k-step pairs, as in conv_wino_kernel
      f32x2 bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float* q = slab + ((kp * 4) * 72 + (lane & 31) + 36 * nt) * 2 + (lane >> 5) * 16 * 72;
        const f32x2 x0 = *reinterpret_cast<const f32x2*>(q), x1 = *reinterpret_cast<const f32x2*>(q + 144),
                    x2 = *reinterpret_cast<const f32x2*>(q + 288), x3 = *reinterpret_cast<const f32x2*>(q + 432);
        bf[nt] = c0 * x0;
        bf[nt] = __builtin_elementwise_fma(c1, x1, bf[nt]);
        bf[nt] = __builtin_elementwise_fma(c2, x2, bf[nt]);
        bf[nt] = __builtin_elementwise_fma(c3, x3, bf[nt]);
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][kp >> 1][2 * (kp & 1) + k2], bf[nt][k2], acc[mt][nt], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * THREADS + tid] = s;
}

// ---- bf16 x 6 form: per k-block of 16: NT x (8 float2 LDS reads, 16 packed transform ops, split8), 6 MT NT MFMAs ----
template <int MT, int NT>
__global__ __launch_bounds__(THREADS) void loop_bf16x6(const unsigned* __restrict__ A3, float* __restrict__ out, int kblocks) {
  __shared__ __attribute__((aligned(16))) float slab[16 * 4 * 72 * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * 4 * 72 * 2; i += THREADS) slab[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 113) - 0.05f;
  __syncthreads();
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const unsigned* ap = A3 + ((size_t)(blockIdx.x % 8) * 12 + wave) * 8192 + lane * 4;
  const f32x2 c0 = {4.f, 4.f}, c1 = {-5.f, -5.f}, c2 = {1.f, 1.f}, c3 = {0.5f, 0.5f};
  for (int kb = 0; kb < kblocks; ++kb) {
    bf16x8 ah[MT], am[MT], al[MT];                 // pre-split weights: 3 x 16 bytes per lane and row tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned* p = ap + ((kb & 7) * MT + mt) * 768;
      ah[mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
      am[mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 256));
      al[mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 512));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v[8];
#pragma unroll
      for (int kp = 0; kp < 4; ++kp) {
        const float* q = slab + ((kp * 4) * 72 + (lane & 31) + 36 * nt) * 2 + (lane >> 5) * 16 * 72;
        const f32x2 x0 = *reinterpret_cast<const f32x2*>(q), x1 = *reinterpret_cast<const f32x2*>(q + 144),
                    x2 = *reinterpret_cast<const f32x2*>(q + 288), x3 = *reinterpret_cast<const f32x2*>(q + 432);
        f32x2 t = c0 * x0;
        t = __builtin_elementwise_fma(c1, x1, t);
        t = __builtin_elementwise_fma(c2, x2, t);
        t = __builtin_elementwise_fma(c3, x3, t);
        v[2 * kp] = t[0];
        v[2 * kp + 1] = t[1];
      }
      bf16x8 bh, bm, bl;
      split8(v, bh, bm, bl);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x16 c = acc[mt][nt];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh, c, 0, 0, 0);     // small terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mt], bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mt], bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh, c, 0, 0, 0);
        acc[mt][nt] = c;
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * THREADS + tid] = s;
}

// ---- numerics: a 32 x 32 x K product both ways against float64 ------------------------------------------------------
__global__ void gemm_check(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Cf, float* __restrict__ Cb, int K) {
  // one wave: C[32][32] = A[32][K] B[K][32]; A row-major, B given transposed [32][K]
  const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
  f32x16 cf, cb;
  for (int r = 0; r < 16; ++r) { cf[r] = 0.f; cb[r] = 0.f; }
  for (int k0 = 0; k0 < K; k0 += 16) {
    float av[8], bv[8];
    for (int e = 0; e < 8; ++e) { av[e] = A[l31 * K + k0 + 8 * lh + e]; bv[e] = B[l31 * K + k0 + 8 * lh + e]; }
    bf16x8 ah, am, al, bh, bm, bl;
    split8(av, ah, am, al);
    split8(bv, bh, bm, bl);
    cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, cb, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, cb, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, cb, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, cb, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, cb, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, cb, 0, 0, 0);
    for (int k = 0; k < 16; k += 2)       // fp32 MFMA: lane half lh supplies k + lh
      cf = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k0 + k + lh], B[l31 * K + k0 + k + lh], cf, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    Cf[row * 32 + l31] = cf[r];
    Cb[row * 32 + l31] = cb[r];
  }
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int kblocks = argc > 1 ? atoi(argv[1]) : 2048;
  const int blocks = 256;
  float *A, *out;
  unsigned* A3;
  CHECK(hipMalloc(&A, 8 * 12 * 4096 * 4 + (1 << 20)));
  CHECK(hipMalloc(&A3, 8 * 12 * 8192 * 4 + (1 << 20)));
  CHECK(hipMalloc(&out, (size_t)blocks * THREADS * 4));
  {
    std::vector<float> h(8 * 12 * 4096 + (1 << 18));
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 31) % 97) - 0.4f;
    CHECK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned> h3(8 * 12 * 8192 + (1 << 18));
    for (size_t i = 0; i < h3.size(); ++i) h3[i] = 0x3c003b80u + (unsigned)((i * 2654435761u) >> 20 & 0x00ff00ffu);
    CHECK(hipMemcpy(A3, h3.data(), h3.size() * 4, hipMemcpyHostToDevice));
  }
#define RUN(MT, NT)                                                                                                        \
  {                                                                                                                       \
    const double flop = 2.0 * 32 * 32 * 16 * MT * NT * (double)kblocks * 12 * blocks;                                     \
    const double t32 = time_ms([&] { hipLaunchKernelGGL((loop_f32<MT, NT>), dim3(blocks), dim3(THREADS), 0, 0, A, out, kblocks); }, 5);      \
    const double t16 = time_ms([&] { hipLaunchKernelGGL((loop_bf16x6<MT, NT>), dim3(blocks), dim3(THREADS), 0, 0, A3, out, kblocks); }, 5);  \
    printf("wave tile %d x %d of 32 x 32:  fp32 MFMA %7.3f ms = %6.1f TFLOP/s   bf16 x 6 %7.3f ms = %6.1f fp32-equivalent TFLOP/s   x %.2f\n",  \
           MT, NT, t32, flop / t32 / 1e9, t16, flop / t16 / 1e9, t32 / t16);                                             \
  }
  RUN(2, 2)
  RUN(3, 1)
  RUN(4, 1)
  RUN(4, 2)
  // numerics
  const int K = 1024;
  std::vector<float> ha(32 * K), hb(32 * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) / 8388608.f - 1.f); };
  for (auto& v : ha) v = rnd() * expf(3.f * rnd());
  for (auto& v : hb) v = rnd() * expf(3.f * rnd());
  float *dA, *dB, *dCf, *dCb;
  CHECK(hipMalloc(&dA, 32 * K * 4)); CHECK(hipMalloc(&dB, 32 * K * 4)); CHECK(hipMalloc(&dCf, 4096)); CHECK(hipMalloc(&dCb, 4096));
  CHECK(hipMemcpy(dA, ha.data(), 32 * K * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, hb.data(), 32 * K * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gemm_check, dim3(1), dim3(64), 0, 0, dA, dB, dCf, dCb, K);
  std::vector<float> cf(1024), cb(1024);
  CHECK(hipMemcpy(cf.data(), dCf, 4096, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(cb.data(), dCb, 4096, hipMemcpyDeviceToHost));
  double ef = 0, eb = 0, scale = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { ref += (double)ha[i * K + k] * hb[j * K + k]; mag += fabs((double)ha[i * K + k] * hb[j * K + k]); }
      ef = fmax(ef, fabs(cf[i * 32 + j] - ref) / mag);
      eb = fmax(eb, fabs(cb[i * 32 + j] - ref) / mag);
      scale = fmax(scale, mag);
    }
  printf("32 x 32 x %d product against float64, max |error| / sum |a b|:  fp32 MFMA %.3e   bf16 x 6 %.3e\n", K, ef, eb);
  return 0;
}
