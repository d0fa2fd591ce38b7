// Shared helpers for the gfx950 kernels of libflowhigh_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/flowhigh_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void fh_set_error(const char* fmt, ...);

// per-device host state (opt-in flags) is kept in arrays of this many ordinals
#define FH_MAX_DEVICES 64

#define FH_CHECK_ARG(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      fh_set_error(__VA_ARGS__);       \
      return FH_E_ARG;                 \
    }                                  \
  } while (0)

#define FH_CHECK_LAUNCH(name)                                            \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      fh_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return FH_E_LAUNCH;                                                \
    }                                                                    \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Wave-uniform values the compiler cannot prove uniform (anything loaded through a selected
// pointer) are pinned to SGPRs with readfirstlane, so that buffer descriptors built from them
// do not get wrapped in waterfall loops.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const float* uni(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

static inline int fh_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Division of a wave-uniform 0 <= n < 2^31 by a launch constant d >= 1 without the ~30-instruction expansion of a run-time
// integer division: the launcher makes (mul, shift) once on the host, the kernel takes floor(n / d) = (mulhi(mul, n) + n) >> shift
// (3 scalar instructions; Granlund-Montgomery round-up form: mul = floor(2^32 (2^shift - d) / d) + 1, shift = ceil(log2 d)).
// The block -> work mappings of the conv kernels are 8-12 such divisions per block, on the scalar unit all waves of a CU share.
struct fh_fastdiv {
  unsigned mul, shift, d;
};
static inline fh_fastdiv fh_make_fastdiv(unsigned d) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  return {(unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1), l, d};
}
__device__ __forceinline__ int fh_div(int n, const fh_fastdiv& f) {
  return (int)((__umulhi(f.mul, (unsigned)n) + (unsigned)n) >> f.shift);
}
__device__ __forceinline__ int fh_mod(int n, int q, const fh_fastdiv& f) { return n - q * (int)f.d; }      // q = fh_div(n, f)

// 64-lane butterfly reductions (wave = 64 on gfx950).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
