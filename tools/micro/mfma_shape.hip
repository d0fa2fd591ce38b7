// fp32 MFMA shape and sustained rate on a full chip: v_mfma_f32_32x32x2_f32 (16 accumulator registers read + written
// per 2048 MACs) against v_mfma_f32_16x16x4_f32 (4 per 1024 MACs: half the accumulator traffic per MAC).  Operands
// CHANGE on every MFMA (rotating set of registers holding random data, refreshed by a cheap VALU op every 64
// iterations), 12 waves per block, one block per CU.  Prints TFLOP/s and the shader clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/mfma_shape tools/micro/mfma_shape.hip && tools/micro/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(768) void mfma_loop(const float* in, float* out, unsigned long long* clk, int iters) {
  float x[8], y[8];
  for (int i = 0; i < 8; ++i) {
    x[i] = in[(threadIdx.x * 16 + i) & 8191];
    y[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 8191];
  }
  float t = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 32) {
    f32x16 a[4] = {};
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u)           // (static register indices: the operand pairing rotates with u)
#pragma unroll
        for (int j = 0; j < 8; ++j)         // 8 MFMAs = 512 cycles, 4 accumulators, all 16 operand registers in turn
          a[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], y[(j + u) & 7], a[j & 3], 0, 0, 0);
      if ((i & 63) == 56) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = a[j] * 0.5f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = -x[j]; y[j] = y[j] * 0.999f; }
      }
    }
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) t += a[j][r];
  } else {
    f32x4 a[16] = {};
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j)        // 16 MFMAs = 512 cycles, 16 accumulators (a 64 x 64 wave tile of 16 x 16 tiles)
          a[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j & 7], y[(j + u) & 7], a[j], 0, 0, 0);
      if ((i & 63) == 56) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = a[j] * 0.5f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = -x[j]; y[j] = y[j] * 0.999f; }
      }
    }
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) t += a[j][r];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (t == 12345.678f) out[0] = t;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = r1 - r0;
  }
}

int main() {
  float *in, *out;
  unsigned long long* clk;
  (void)hipMalloc(&in, 8192 * 4);
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 20000;
  const int long_iters = 3000000;      // ~2 s: long enough for the chip's power management to settle
  std::vector<float> h(8192);
  for (int mode = 0; mode < 2; ++mode) {
    for (auto& v : h) v = mode ? (float)rand() / RAND_MAX * 2.f - 1.f : 1e-3f;
    (void)hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int shape : {32, 16, 32, 16}) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (shape == 32) mfma_loop<32><<<256, 768>>>(in, out, clk, rep ? iters : 200);
        else mfma_loop<16><<<256, 768>>>(in, out, clk, rep ? iters : 200);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[2];
      (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      const double flop = 256.0 * 12 * iters * 8 * 2.0 * 32 * 32 * 2;        // same MACs per iteration in both shapes
      printf("%s operands, %s: %8.3f ms  %7.1f TFLOP/s   clock %.0f MHz\n", mode ? "random  " : "constant",
             shape == 32 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", ms, flop / ms * 1e-9,
             100.0 * (double)c[0] / (double)c[1]);
    }
  }
  // sustained: the same loop for ~2 s
  for (int shape : {32, 16}) {
    (void)hipEventRecord(e0);
    if (shape == 32) mfma_loop<32><<<256, 768>>>(in, out, clk, long_iters);
    else mfma_loop<16><<<256, 768>>>(in, out, clk, long_iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2];
    (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double flop = 256.0 * 12 * long_iters * 8 * 2.0 * 32 * 32 * 2;
    printf("random operands, sustained, %s: %8.1f ms  %7.1f TFLOP/s   clock %.0f MHz\n",
           shape == 32 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", ms, flop / ms * 1e-9, 100.0 * (double)c[0] / (double)c[1]);
  }
  return 0;
}
