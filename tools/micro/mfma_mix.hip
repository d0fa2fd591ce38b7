// What takes the clock (and the matrix-pipe share) away from an fp32 MFMA loop?  The Winograd K loop holds 2.07-2.15 GHz and
// ~0.78 of the pipe; a register-only loop holds 2.39 GHz and 0.98.  This adds the K loop's other activity to the
// register-only loop one ingredient at a time (12 waves per block, one block per CU, 8 MFMAs per iteration and wave =
// one k-step pair of the 64 x 512 tile):
//   L: 4 ds_read2st64_b64 per iteration (the B^T samples), results feed the MFMA operands
//   V: 6 v_pk_fma_f32 per iteration (the transform)
//   G: 1 16-byte global load per 2 iterations (the weight fragments, from a small L2-resident buffer)
//   B: a block barrier every 16 iterations (one per chunk of 4 tap groups)
//   H: a 16-byte global load per 8 iterations from a 1 GB buffer, every wave its own stream: ~1.4 TB/s of HBM reads
//   W: a 16-byte global store per 16 iterations to a 1 GB buffer: ~0.7 TB/s of HBM writes
//   S: 8 ds_write_b64 per 16 iterations (the slab staging of one chunk of 4 tap groups)
//   A: 14 scalar ALU instructions per iteration (the real loop's descriptor / address arithmetic: 1.8 SALU per MFMA, PMC)
//   P: the weight loads walk a 64 MB buffer shared by all blocks instead of an L2-resident 4 KB tile (L2 misses to the Infinity Cache)
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/mfma_mix tools/micro/mfma_mix.hip && tools/micro/mfma_mix [random]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MASK>
__global__ __launch_bounds__(768) void mix_loop(const float* __restrict__ in, float* out, unsigned long long* clk, int iters,
                                                 const f32x4* __restrict__ big, f32x4* __restrict__ bigw) {
  constexpr bool L = MASK & 1, V = MASK & 2, G = MASK & 4, B = MASK & 8, H = MASK & 16, W = MASK & 32, S = MASK & 64, P = MASK & 128, A = MASK & 256;
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 768) lds[i] = in[i & 8191];
  __syncthreads();
  f32x16 a[4] = {};
  float x[8];
  f32x2 bf[2] = {{in[tid & 8191], in[(tid + 1) & 8191]}, {in[(tid + 2) & 8191], in[(tid + 3) & 8191]}};
  for (int i = 0; i < 8; ++i) x[i] = in[(tid * 16 + i) & 8191];
  const f32x2 c0 = {-4.f, -4.f}, c1 = {-1.f, -1.f}, c2 = {1.f, 1.f};
  const float* lp = lds + (tid >> 6) * 1024 + (lane & 31) * 2;
  const f32x4* gp = reinterpret_cast<const f32x4*>(in) + (lane & 31) * 4 + (lane >> 5);
  // 1 GB = 2^26 f32x4; every wave walks its own 64-lane-wide stream
  const size_t hbase = ((size_t)blockIdx.x * 12 + (tid >> 6)) * 64 + lane;
  f32x4 hacc = {0.f, 0.f, 0.f, 0.f};
  int sacc = __builtin_amdgcn_readfirstlane(tid);
  const int i_dummy = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i0 = 0; i0 < iters; i0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {          // (two iterations unrolled: static register indices everywhere)
      const int i = i0 + u;
      f32x2 s[4][2];
      if (L) {
        const float* q0 = lp + ((i0 >> 1) & 7) * 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* q = q0 + ((u + r) & 3) * 272;
          s[r][0] = *reinterpret_cast<const f32x2*>(q);
          s[r][1] = *reinterpret_cast<const f32x2*>(q + 128);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r][0] = bf[0]; s[r][1] = bf[1]; }
      }
      if (V) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x2 p = __builtin_elementwise_fma(c0, s[0][nt], s[1][nt]);
          const f32x2 q = __builtin_elementwise_fma(c1, s[2][nt], s[3][nt]);
          bf[nt] = __builtin_elementwise_fma(c2, q, p) * 0.25f;
        }
      } else if (L) {
        bf[0] = s[0][0]; bf[1] = s[1][1];
        asm volatile("" :: "v"(s[2][0]), "v"(s[3][0]), "v"(s[0][1]), "v"(s[2][1]), "v"(s[3][1]), "v"(s[1][0]));
      }
      if (G && u == 1) {
        const f32x4 w = P ? big[((size_t)(i0 >> 1) * 64 + lane) & ((1ull << 22) - 1)] : gp[(i0 & 62) * 8];
        x[0] = w[0]; x[1] = w[1]; x[2] = w[2]; x[3] = w[3];
      }
      if (A) {
#pragma unroll
        for (int e = 0; e < 7; ++e) asm volatile("s_mul_i32 %0, %0, 3\n\ts_add_u32 %0, %0, 1" : "+s"(sacc));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          a[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[(j & 1) * 2 + k2 + 4 * u], bf[j >> 1][k2], a[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (S && (i0 & 15) == 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        *reinterpret_cast<f32x2*>(lds + 8192 + ((tid * 2 + e * 1536) & 8191)) = bf[e & 1];
    }
    if (H && (i0 & 7) == 6) hacc += big[(hbase + (size_t)(i0 >> 3) * (256 * 12 * 64)) & ((1ull << 26) - 1)];
    if (W && (i0 & 15) == 14) bigw[(hbase + (size_t)(i0 >> 4) * (256 * 12 * 64)) & ((1ull << 26) - 1)] = hacc;
    if (B && (i0 & 15) == 14) __syncthreads();
    if ((i0 & 255) == 254) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = a[j] * 0.001f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float t = hacc[0] + hacc[1] + hacc[2] + hacc[3] + (float)sacc;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) t += a[j][r];
  if (t == 12345.678f) out[0] = t;
  if (threadIdx.x == 0 && blockIdx.x == 17) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
  (void)i_dummy;
}

template <int MASK>
void run(const float* in, float* out, unsigned long long* clk, const char* name, const f32x4* big, f32x4* bigw) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 400000;       // ~0.3 s
  mix_loop<MASK><<<256, 768>>>(in, out, clk, 1000, big, bigw);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mix_loop<MASK><<<256, 768>>>(in, out, clk, iters, big, bigw);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[2];
  (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  const double flop = 256.0 * 12 * iters * 8 * 2.0 * 32 * 32 * 2;
  const double mhz = 100.0 * (double)c[0] / (double)c[1];
  printf("%-28s %8.1f ms  %6.1f TFLOP/s  clock %4.0f MHz  matrix-pipe share %.3f\n", name, ms, flop / ms * 1e-9, mhz,
         (3.0 * iters * 8 * 64) / (ms * 1e-3 * mhz * 1e6));
}

// Many short launches back to back (the real conv launches are ~300 us): does the clock / rate of a launch depend on its length?
template <int MASK>
void run_short(const float* in, float* out, unsigned long long* clk, const char* name, const f32x4* big, f32x4* bigw, int iters, int launches) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int l = 0; l < 20; ++l) mix_loop<MASK><<<256, 768>>>(in, out, clk, iters, big, bigw);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int l = 0; l < launches; ++l) mix_loop<MASK><<<256, 768>>>(in, out, clk, iters, big, bigw);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[2];
  (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  const double flop = 256.0 * 12 * iters * 8 * 2.0 * 32 * 32 * 2 * launches;
  printf("%-28s %5d launches of %6d iterations (%7.1f us each): %6.1f TFLOP/s over all, clock of the last launch %4.0f MHz\n", name,
         launches, iters, ms * 1e3 / launches, flop / ms * 1e-9, 100.0 * (double)c[0] / (double)c[1]);
}

int main(int argc, char** argv) {
  const bool random_big = argc > 1;       // any argument: the 1 GB weight / stream buffer holds random values instead of zeros
  float *in, *out;
  unsigned long long* clk;
  (void)hipMalloc(&in, 65536 * 4);
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&clk, 16);
  std::vector<float> h(65536);
  for (auto& v : h) v = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
  (void)hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  f32x4 *big, *bigw;
  (void)hipMalloc(&big, 1ull << 30);
  (void)hipMalloc(&bigw, 1ull << 30);
  (void)hipMemset(big, 0, 1ull << 30);
  if (random_big) {                       // (data-dependent power: the same loops on operands whose bits toggle)
    std::vector<float> hb(1u << 24);
    for (auto& v : hb) v = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    for (size_t off = 0; off < (1ull << 30); off += hb.size() * 4)
      (void)hipMemcpy(reinterpret_cast<char*>(big) + off, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    printf("weights / streamed operands: random in [-1, 1]\n");
  }
  run<0>(in, out, clk, "MFMA only", big, bigw);
  run<1>(in, out, clk, "+ LDS reads", big, bigw);
  run<2>(in, out, clk, "+ packed transform", big, bigw);
  run<4>(in, out, clk, "+ weight loads", big, bigw);
  run<8>(in, out, clk, "+ barrier per 16", big, bigw);
  run<15>(in, out, clk, "+ all four (L V G B)", big, bigw);
  run<16>(in, out, clk, "MFMA + HBM reads", big, bigw);
  run<48>(in, out, clk, "MFMA + HBM reads + writes", big, bigw);
  run<63>(in, out, clk, "all six", big, bigw);
  run<64>(in, out, clk, "MFMA + LDS slab writes", big, bigw);
  run<4 + 128>(in, out, clk, "MFMA + weight loads (64 MB)", big, bigw);
  run<63 + 64 + 128>(in, out, clk, "all eight", big, bigw);
  run<256>(in, out, clk, "MFMA + 14 SALU", big, bigw);
  run<63 + 64 + 128 + 256>(in, out, clk, "all nine", big, bigw);
  run<0>(in, out, clk, "MFMA only (again)", big, bigw);
  for (int iters : {100000, 10000, 1000, 460, 100}) {
    run_short<0>(in, out, clk, "MFMA only", big, bigw, iters, 40000000 / iters > 2000 ? 2000 : 40000000 / iters);
    run_short<63 + 64 + 128 + 256>(in, out, clk, "all nine", big, bigw, iters, 40000000 / iters > 2000 ? 2000 : 40000000 / iters);
  }
  return 0;
}
