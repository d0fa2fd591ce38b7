// Does the fp32 MFMA rate per CU depend on how many CUs are busy and on the data (power / clock management)?
// Pure-register MFMA loop, 12 waves per block, one block per CU; timed with 32 ... 1024 blocks, with
// constant and with random operands; the shader clock is read as d(s_memtime) / d(s_memrealtime @ 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(768) void mfma_loop(const float* in, float* out, unsigned long long* clk, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x[4], y[4];
  for (int i = 0; i < 4; ++i) {
    x[i] = in[(threadIdx.x * 8 + i) & 8191];
    y[i] = in[(threadIdx.x * 8 + 4 + i + blockIdx.x) & 8191];
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[0], y[0], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[1], y[1], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[2], y[2], a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[3], y[3], a3, 0, 0, 0);
    if ((i & 63) == 63) {          // keep the accumulators bounded but busy
      a0 = a0 * 0.5f - a2 * 0.25f;
      a1 = a1 * 0.5f - a3 * 0.25f;
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x16 s = a0 + a1 + a2 + a3;
  float t = 0;
  for (int i = 0; i < 16; ++i) t += s[i];
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
  const int iters = 40000;
  std::vector<float> h(8192);
  for (int mode = 0; mode < 2; ++mode) {
    for (auto& v : h) v = mode ? (float)rand() / RAND_MAX * 2.f - 1.f : 1e-3f;
    (void)hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int blocks : {32, 128, 256, 1024}) {
      mfma_loop<<<blocks, 768>>>(in, out, clk, 100);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      mfma_loop<<<blocks, 768>>>(in, out, clk, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[2];
      (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      const double flop = (double)blocks * 12 * iters * 4 * 2.0 * 32 * 32 * 2;
      printf("%s operands, blocks %4d: %8.3f ms  %7.1f TFLOP/s   s_memtime/s_memrealtime = %.3f (x 100 MHz)\n",
             mode ? "random  " : "constant", blocks, ms, flop / ms * 1e-9, (double)c[0] / (double)c[1]);
    }
  }
  return 0;
}
