// Issue rate of v_fma_f32 / v_pk_fma_f32 / v_rndne_f32 / ds_read_b128 on one SIMD (cycles per wave64 instruction),
// with 1, 2 and 4 waves per SIMD.  One block of 256 x W threads per CU; shader clock from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void rate_kernel(float* out, unsigned long long* clk, int iters) {
  f32x2 a[8];
  float s[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = (f32x2){threadIdx.x * 1e-3f + i, 1.f};
    s[i] = threadIdx.x * 1e-3f + i;
  }
  const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m[0]), "v"(c[0]));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (MODE == 2) asm volatile("v_rndne_f32 %0, %0" : "+v"(s[i]));
      if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(m), "v"(c));   // SGPR constant
      if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(a[(i + 3) & 7]), "v"(m));   // broadcast low half
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float t = 0;
  for (int i = 0; i < 8; ++i) t += a[i][0] + a[i][1] + s[i];
  if (t == 12345.f) out[0] = t;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* clk) {
  const int iters = 200000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w : {1, 2, 4, 6, 8}) {
    const int blocks = w > 4 ? 2 : 1, per = w / blocks;       // 6, 8 waves per SIMD: two co-resident blocks per CU
    rate_kernel<MODE><<<256 * blocks, 256 * per>>>(out, clk, 1000);
    (void)hipEventRecord(e0);
    rate_kernel<MODE><<<256 * blocks, 256 * per>>>(out, clk, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    // (block 0's own cycle count, and the launch's wall time: with two blocks per CU the second may run after the first)
    printf("%-14s %d wave(s)/SIMD: %.2f cycles per instruction per SIMD by s_memtime of block 0, %.2f ns per instruction per SIMD by events\n",
           name, w, (double)c / (8.0 * iters * w), ms * 1e6 / (8.0 * iters * w));
  }
}

int main() {
  float* out;
  unsigned long long* clk;
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&clk, 8);
  run<0>("v_fma_f32", out, clk);
  run<1>("v_pk_fma_f32", out, clk);
  run<2>("v_rndne_f32", out, clk);
  run<3>("v_pk_mul_f32", out, clk);
  run<4>("pk_fma sgpr", out, clk);
  run<5>("pk_fma bcast", out, clk);
  return 0;
}
