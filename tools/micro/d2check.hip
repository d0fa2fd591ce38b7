// Does v_dot2c_f32_bf16 with the constant pair {-1, 0} / {0, -1} give the EXACT remainder x - bf16(x) of the bf16 x 6 split
// (flowhigh_amd/csrc/conv_wino54_kernel.h: v_split8)?  Compares it with the subtraction route on n random bit patterns per class.
//   hipcc --offload-arch=gfx950 -O3 -o d2check d2check.hip && ./d2check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, int n, unsigned long long* bad, unsigned long long* bad_flush, float* ex) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  const bf2 h = {(__bf16)a, (__bf16)b};
  const unsigned hp = __builtin_bit_cast(unsigned, h);
  const bf2 slo = __builtin_bit_cast(bf2, 0x0000BF80u), shi = __builtin_bit_cast(bf2, 0xBF800000u);
  const float ra = __builtin_amdgcn_fdot2_f32_bf16(h, slo, a, false), rb = __builtin_amdgcn_fdot2_f32_bf16(h, shi, b, false);
  const float sa = a - __uint_as_float(hp << 16), sb = b - __uint_as_float(hp & 0xffff0000u);
  const bool ea = __float_as_uint(ra) != __float_as_uint(sa) && !(ra == 0.f && sa == 0.f);
  const bool eb = __float_as_uint(rb) != __float_as_uint(sb) && !(rb == 0.f && sb == 0.f);
  if (ea || eb) {
    const float r = ea ? ra : rb, s = ea ? sa : sb;
    const bool flush = r == 0.f && fabsf(s) < 1.17549435e-38f;       // the dot unit flushes a denormal result
    atomicAdd(flush ? bad_flush : bad, 1ull);
    if (!flush) { ex[0] = ea ? a : b; ex[1] = r; ex[2] = s; }
  }
}
int main() {
  const int n = 1 << 26;
  std::vector<float> h(n);
  float* d; unsigned long long *bad; float* ex;
  hipMalloc(&d, n * 4); hipMalloc(&bad, 16); hipMalloc(&ex, 12);
  const char* names[] = {"random bit patterns", "normal range 1e-6..1e3", "tiny 2^-130..2^-110", "second-level remainders (x - bf16(x))"};
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int cls = 0; cls < 4; ++cls) {
    for (int i = 0; i < n; ++i) {
      uint32_t u = (uint32_t)rnd();
      float f;
      if (cls == 0) { if (((u >> 23) & 255) == 255) u &= ~(1u << 30); memcpy(&f, &u, 4); }
      else if (cls == 1) { u = (u & 0x807fffffu) | ((107 + (rnd() % 30)) << 23); memcpy(&f, &u, 4); }
      else if (cls == 2) { u = (u & 0x807fffffu) | ((uint32_t)(rnd() % 18) << 23); memcpy(&f, &u, 4); }
      else { u = (u & 0x807fffffu) | ((107 + (rnd() % 30)) << 23); memcpy(&f, &u, 4);
             uint32_t t = u + 0x7fff + ((u >> 16) & 1); t &= 0xffff0000u; float hf; memcpy(&hf, &t, 4); f = f - hf; }
      h[i] = f;
    }
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 16); hipMemset(ex, 0, 12);
    k<<<n / 2 / 256, 256>>>(d, n, bad, bad + 1, ex);
    unsigned long long hb[2]; float he[3];
    hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 12, hipMemcpyDeviceToHost);
    printf("%-40s n = %d: %llu differ, %llu more differ only by a flushed denormal remainder", names[cls], n, hb[0], hb[1]);
    if (hb[0]) printf("   e.g. x = %a: dot2c %a, sub %a", he[0], he[1], he[2]);
    printf("\n");
  }
  return 0;
}
