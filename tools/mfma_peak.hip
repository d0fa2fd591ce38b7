// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate (register operands only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD > 1024 ? 1024 : 256 * WAVES_PER_SIMD) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int W> void run(const char* name) {
  const int threads = 256 * W > 1024 ? 1024 : 256 * W;
  const int blocks = 256 * (256 * W / threads);
  float* out; hipMalloc(&out, sizeof(float) * blocks * threads);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(threads), 0, 0, out, 100, 0.5f, 0.25f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (threads / 64) * iters * 32.0 * 4096.0;
    printf("%s: %.2f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
  }
  hipFree(out);
}
int main() { run<1>("1 wave/SIMD"); run<2>("2 waves/SIMD"); run<4>("4 waves/SIMD"); return 0; }
