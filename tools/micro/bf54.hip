// Micro-benchmark (round 6): the K loop an F(5,4) bf16 x 6 wide-stage kernel would run, with its ingredients switchable.
//
// One block per CU, 8 waves = 8 transform points (2 per SIMD), wave tile (32 MT) rows x 2 columns of 32 tiles.  Per tap group
// (16-channel k-block) and column ("item"): the lane's 8 channels of 6 samples from the LDS slab, a 5-FMA chain per value
// (one row of B^T), the exact split into three bf16 pieces, 6 MT v_mfma_f32_32x32x16_bf16.  Weights: three 16-byte pieces per
// lane and row tile from global memory (L2), per chunk a slab staging (4 x 16-byte global loads, LDS writes, one barrier).
//
//   FLAGS (template MASK):
//     1 A2     weights double-buffered in registers: group g + 1 is requested before the MFMAs of group g (else: behind them, one set)
//     2 PIPE   software pipeline inside a wave: item i + 1 is prepared between the MFMAs of item i (sched_group_barrier)
//     4 NOA    no weight loads            8 NOLDS  no LDS reads (samples stay in registers)
//    16 NOXF   no transform              32 NOSPLIT no split (raw bits as pieces)
//    64 NOSTAGE no slab staging / barrier 128 NOMFMA no matrix instructions
//   256 PK     transform and split in packed fp32 (v_pk_fma_f32 / v_pk_add_f32), as round 5's kernel had them
//   512 CNT    the per-chunk block barrier replaced by two LDS counters per slab buffer (written / read by 8 waves): a wave waits
//              for data, never for the other waves' arithmetic -- the waves of a SIMD stay out of phase (one in its MFMAs while the
//              other prepares its next item)
//  4096 XLATE  the slab's global loads are issued BEHIND the chunk's first weight request (vmcnt retires in order: a wait for weights
//              must not sit behind an HBM round trip)     8192 X2  ... and two chunks ahead (two register sets)
// 16384 PIN    a scheduling barrier behind the global loads (the compiler otherwise sinks the slab's loads to their use at the end
//              of the chunk to save registers, and the whole HBM round trip is waited out there)
// 65536 ALAY   weights stored in fragment order, [row tile][piece][lane][16 bytes]: a wave's 16-byte load is 1 KB contiguous (8 cache
//              lines) instead of 32 rows x 2 x 16 bytes at a 96-byte pitch (24 lines)
//  1024 NOBAR  no synchronisation at all (racy: timing only)     2048 PRIO  the two waves of a SIMD run at different priorities
//
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -o tools/micro/bf54 tools/micro/bf54.hip && tools/micro/bf54
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 512;
constexpr int P = 72;                     // tiles per (plane, channel quad) row
constexpr int QROW = 4 * P;               // floats of one channel quad inside a plane
constexpr int PLANE = 4 * QROW;           // floats of a plane (16 channels)
constexpr int SLAB = 5 * PLANE;           // 5 planes

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(__builtin_amdgcn_perm(0u, p, 0x01000c0cu)); }

template <bool PK>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    h[i] = pack_bf16(a, b);
    float ra, rb, sa, sb;
    if (PK) {
      f32x2 r = (f32x2){a, b} - (f32x2){__uint_as_float(h[i] << 16), __uint_as_float(h[i] & 0xffff0000u)};
      m[i] = pack_bf16(r[0], r[1]);
      f32x2 s = r - (f32x2){__uint_as_float(m[i] << 16), __uint_as_float(m[i] & 0xffff0000u)};
      sa = s[0]; sb = s[1];
    } else {
      ra = a - bf_lo(h[i]); rb = b - __uint_as_float(h[i] & 0xffff0000u);
      m[i] = pack_bf16(ra, rb);
      sa = ra - bf_lo(m[i]); sb = rb - __uint_as_float(m[i] & 0xffff0000u);
    }
    l[i] = pack_bf16(sa, sb);
  }
}

template <int MT, int MASK, int GC>
__global__ __launch_bounds__(THREADS, 1) __attribute__((amdgpu_waves_per_eu(2, 2)))
void loop54(const float* __restrict__ A3, const float* __restrict__ X, float* __restrict__ out, unsigned long long* clk, int chunks,
            int a_panel_floats) {
  constexpr bool A2 = MASK & 1, PIPE = MASK & 2, NOA = MASK & 4, NOLDS = MASK & 8, NOXF = MASK & 16, NOSPLIT = MASK & 32,
                 NOSTAGE = MASK & 64, NOMFMA = MASK & 128, PK = MASK & 256, CNT = MASK & 512, NOBAR = MASK & 1024, PRIO = MASK & 2048, XLATE = MASK & 4096, X2 = MASK & 8192, PIN = MASK & 16384, TRACE = MASK & 32768, ALAY = MASK & 65536;
  extern __shared__ __attribute__((aligned(16))) float lds[];      // 2 slabs | 4 counters
  unsigned* const cnt = reinterpret_cast<unsigned*>(lds + 2 * SLAB);      // [0..1] waves that wrote buffer b, [2..3] waves done reading b
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * SLAB; i += THREADS) lds[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 113) - 0.05f;
  if (tid < 4) cnt[tid] = tid == 0 ? 8u : 0u;                 // buffer 0 holds chunk 0
  __syncthreads();
  if (PRIO) { if (xi >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
  // counter synchronisation (CNT): monotonic counters, targets = 8 x (uses so far)
  auto wait_cnt = [&](int idx, unsigned target) {
    while (true) {
      const unsigned v = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(cnt + idx));
      if ((int)(v - target) >= 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
  };
  auto bump_cnt = [&](int idx) {
    __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): this wave's LDS accesses are done
    if (lane == 0) __hip_atomic_fetch_add(cnt + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // end of chunk `c` (running index): publish the next chunk's slab part, release this chunk's buffer, wait for the next one
  auto chunk_sync = [&](int c, int xbuf) {
    if (NOSTAGE || NOBAR) return;
    if (!CNT) { __syncthreads(); return; }
    bump_cnt(xbuf ^ 1);                                        // my part of chunk c + 1 is written
    bump_cnt(2 + xbuf);                                        // I am done reading chunk c
    wait_cnt(xbuf ^ 1, 8u * (unsigned)((c + 1) / 2 + 1));      // all parts of chunk c + 1 are there
  };
  // before writing chunk c + 1 into the other buffer: every wave must be done reading chunk c - 1 there
  auto wait_free = [&](int c, int xbuf) {
    if (NOSTAGE || NOBAR || !CNT) return;
    wait_cnt(2 + (xbuf ^ 1), 8u * (unsigned)((c + 1) / 2));
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this wave's row of B^T (wave-uniform coefficients) and sample offsets (floats): sample c -> plane c % 5, index c / 5
  float co[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) co[j] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(0.25f * (float)(j + 1) - 0.5f * (float)(xi & 3))));
  const int lane_base = (2 * lh) * QROW + l31 * 4;

  // weights: [chunk][g][xi][row 32 MT][3 pieces][16 bf16] = 24 floats per row; lane (row l31, half lh) reads 16 bytes per piece
  const int a_lane = (l31 * 24 + lh * 4) * 4;
  const float* a_base = A3 + (size_t)(blockIdx.x & 7) * a_panel_floats;       // 8 panels, shared by the blocks of an "XCD"
  u32x4 a3[A2 ? 2 : 1][MT][3];
  auto load_a = [&](int set, int grp) {       // grp: flat (chunk, g) index
    if (NOA) return;
    const float* up = a_base + ((size_t)grp * 8 + xi) * (32 * MT * 24);
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, 32 * MT * 24 * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        a3[set][mt][pc] = ALAY ? __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + (mt * 3 + pc) * 1024, 0, 0)
                               : __builtin_amdgcn_raw_buffer_load_b128(r, a_lane + mt * 32 * 24 * 4 + 32 * pc, 0, 0);
  };
  if (NOA) {
#pragma unroll
    for (int s = 0; s < (A2 ? 2 : 1); ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) a3[s][mt][pc] = (u32x4){0x3c003b80u + tid, 0x3c013b81u, 0x3b803c00u + mt, 0x3c003b80u + pc};
  }

  // slab staging: wave w = (quad q = w & 3, half of the samples w >> 2): lane loads 4 samples of 4 channel rows, writes 4 b128
  u32x4 xqs[2][4];
  auto stage_load = [&](int chunk, int xs = 0) {
    u32x4 (&xq)[4] = xqs[xs];
    if (NOSTAGE) return;
    const float* xp = X + ((size_t)((blockIdx.x >> 3) * 64 + (chunk & 63)) * 16 + 4 * (xi & 3)) * 512 + (xi >> 2) * 256 + lane * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) xq[r] = *reinterpret_cast<const u32x4*>(xp + r * 512);
  };
  int wofs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int w = ((xi >> 2) * 64 + lane) * 4 + e;                 // local sample
    const int wc = w < 5 * P ? w : 0;
    wofs[e] = (wc % 5) * PLANE + (xi & 3) * QROW + (wc / 5) * 4;
  }
  auto stage_store = [&](int buf, int xs = 0) {
    u32x4 (&xq)[4] = xqs[xs];
    if (NOSTAGE) return;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      *reinterpret_cast<u32x4*>(lds + buf * SLAB + wofs[e]) = (u32x4){xq[0][e], xq[1][e], xq[2][e], xq[3][e]};
  };

  // ---- one item: prepare (samples -> transform -> split) and multiply -------------------------------------------------------
  float xkeep[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) xkeep[e] = 0.01f * (float)(tid + e);
  auto prep = [&](int xbuf, int g, int nt, u32x4 (&pc)[3]) {
    float v[8];
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {                 // the lane's two channel quads
      f32x4 x[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int c = 4 * g + j + 1;
        if (NOLDS) {
          x[j] = (f32x4){xkeep[(j + hq) & 7], xkeep[(j + 1) & 7], xkeep[(j + 2) & 7], xkeep[(j + 3 + hq) & 7]};
        } else {
          x[j] = *reinterpret_cast<const f32x4*>(lds + xbuf * SLAB + lane_base + hq * QROW + (c % 5) * PLANE + (c / 5) * 4 + nt * 128);
        }
      }
      if (NOXF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * hq + e] = x[(e + hq) % 6][e];
        asm volatile("" :: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]));
      } else if (PK) {
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          f32x2 t = {x[5][2 * e2], x[5][2 * e2 + 1]};
#pragma unroll
          for (int j = 0; j < 5; ++j) t = __builtin_elementwise_fma((f32x2){co[j], co[j]}, (f32x2){x[j][2 * e2], x[j][2 * e2 + 1]}, t);
          v[4 * hq + 2 * e2] = t[0];
          v[4 * hq + 2 * e2 + 1] = t[1];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = x[5][e];
#pragma unroll
          for (int j = 0; j < 5; ++j) t = __builtin_fmaf(co[j], x[j][e], t);
          v[4 * hq + e] = t;
        }
      }
    }
    if (NOSPLIT) {
      pc[0] = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
      pc[1] = (u32x4){__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
      pc[2] = pc[0] ^ pc[1];
    } else {
      split8<PK>(v, pc[0], pc[1], pc[2]);
    }
  };
  auto mul = [&](int set, int nt, const u32x4 (&pc)[3]) {
    if (NOMFMA) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][nt][e] += __uint_as_float(pc[0][e] ^ pc[1][e] ^ pc[2][e] ^ a3[set][mt][0][e] ^ a3[set][mt][1][e] ^ a3[set][mt][2][e]);
      return;
    }
    const bf16x8 bh = __builtin_bit_cast(bf16x8, pc[0]), bm = __builtin_bit_cast(bf16x8, pc[1]), bl = __builtin_bit_cast(bf16x8, pc[2]);
    // piece pair order (small terms first); all row tiles per pair: four independent accumulators between dependent MFMAs
#pragma unroll
    for (int pp = 0; pp < 6; ++pp) {
      constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, a3[set][mt][pa[pp]]);
        const bf16x8 b = pb[pp] == 0 ? bh : pb[pp] == 1 ? bm : bl;
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[mt][nt], 0, 0, 0);
      }
    }
  };

  load_a(0, 0);
  unsigned long long tr_prep = 0, tr_mul = 0, tr_sync = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int xbuf = 0, grp = 0;
  if constexpr (!PIPE) {
    if (X2) stage_load(1, 1);
    for (int c = 0; c < chunks; c += 2) {            // two chunks per trip: static register parities for any GC
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (!XLATE && !X2) stage_load(c + cc + 1);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GC; ++g) {
          const int set = A2 ? ((cc * GC + g) & 1) : 0;
          if (A2) load_a(set ^ 1, grp + 1);
          if (g == 0 && A2 && XLATE) stage_load(c + cc + 1);
          if (g == 0 && A2 && X2) stage_load(c + cc + 2, cc);          // chunk c + cc + 1 sits in set cc ^ 1
          if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            u32x4 pc[3];
            unsigned long long ta = 0, tb = 0, tc = 0;
            if (TRACE) { __builtin_amdgcn_sched_barrier(0); ta = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            prep(xbuf, g, nt, pc);
            if (TRACE) { __builtin_amdgcn_sched_barrier(0); asm volatile("" :: "v"(pc[0]), "v"(pc[1]), "v"(pc[2])); tb = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            mul(set, nt, pc);
            if (TRACE) { __builtin_amdgcn_sched_barrier(0); tc = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); tr_prep += tb - ta; tr_mul += tc - tb; }
          }
          if (!A2) load_a(0, grp + 1);
          if (g == 0 && !A2 && XLATE) stage_load(c + cc + 1);
          ++grp;
        }
        unsigned long long td = 0;
        if (TRACE) { __builtin_amdgcn_sched_barrier(0); td = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        wait_free(c + cc, xbuf);
        stage_store(xbuf ^ 1, X2 ? cc ^ 1 : 0);
        chunk_sync(c + cc, xbuf);
        if (TRACE) { __builtin_amdgcn_sched_barrier(0); tr_sync += __builtin_amdgcn_s_memtime() - td; __builtin_amdgcn_sched_barrier(0); }
        xbuf ^= 1;
      }
    }
  } else {
    // software pipeline over the flat item sequence (chunk, g, nt); pieces of item i + 1 are made between the MFMAs of item i
    u32x4 pcs[2][3];
    prep(0, 0, 0, pcs[0]);
    for (int c = 0; c < chunks; c += 2) {            // two chunks per trip: static register parities for any GC
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        stage_load(c + cc + 1);
#pragma unroll
        for (int g = 0; g < GC; ++g) {
          constexpr int dummy = 0;
          const int fl = cc * GC + g;                // flat group index within the trip (static)
          const int set = A2 ? (fl & 1) : 0;
          if (A2) load_a(set ^ 1, grp + 1);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int it = (fl * 2 + nt) & 1;        // piece set parity (static)
            // next item: (g, 1) | (g + 1, 0) | next chunk's (0, 0) in the other slab buffer (staged + barrier below)
            const bool last_of_chunk = g == GC - 1 && nt == 1;
            if (!last_of_chunk) {
              prep(xbuf, nt == 0 ? g : g + 1, nt ^ 1, pcs[it ^ 1]);
              mul(set, nt, pcs[it]);
              // interleave request: one MFMA, then a share of the item's vector / LDS work
#pragma unroll
              for (int k = 0; k < 6 * MT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
                if (k < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // 1 LDS read
                __builtin_amdgcn_sched_group_barrier(0x002, (84 + 6 * MT - 1) / (6 * MT), 0);   // VALU share
              }
            } else {
              mul(set, nt, pcs[it]);
              if (!A2) load_a(0, grp + 1);
              wait_free(c + cc, xbuf);
              stage_store(xbuf ^ 1);
              chunk_sync(c + cc, xbuf);
              xbuf ^= 1;
              prep(xbuf, 0, 0, pcs[it ^ 1]);
            }
          }
          if (!A2 && g < GC - 1) load_a(0, grp + 1);
          ++grp;
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * THREADS + tid] = s + __uint_as_float((xqs[0][0][0] ^ xqs[1][0][0]) & 1u);
  if (tid == 0) { clk[blockIdx.x] = t1 - t0; clk[256 + 256 * 6 + blockIdx.x] = r1 - r0; }
  if (TRACE && lane == 0 && (xi == 0 || xi == 7)) {
    unsigned long long* q = clk + 256 + (blockIdx.x * 2 + (xi == 7)) * 3;
    q[0] = tr_prep; q[1] = tr_mul; q[2] = tr_sync;
  }
}

static float *dA, *dX, *dOut;
static unsigned long long* dClk;

template <typename F>
static double time_ms(F launch, int reps);

// ---- the same loop with SIXTEEN waves per block: wave = (point, tile column), 4 waves per SIMD at <= 128 registers -------------------------
// (does the loop's latency chain -- LDS round trips, dependent FMA / split chains, L2 waits -- hide under twice the waves?)
template <int MT, int GC>
__global__ __launch_bounds__(1024, 1) __attribute__((amdgpu_waves_per_eu(4, 4)))
void loop54w16(const float* __restrict__ A3, const float* __restrict__ X, float* __restrict__ out, int chunks, int a_panel_floats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave & 7, nt = wave >> 3;
  for (int i = tid; i < 2 * SLAB; i += 1024) lds[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 113) - 0.05f;
  __syncthreads();
  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float co[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) co[j] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(0.25f * (float)(j + 1) - 0.5f * (float)(xi & 3))));
  const int lane_base = (2 * lh) * QROW + l31 * 4 + nt * 128;
  const float* a_base = A3 + (size_t)(blockIdx.x & 7) * a_panel_floats;
  u32x4 a3[MT][3];
  auto load_a = [&](int grp) {
    const float* up = a_base + ((size_t)grp * 8 + xi) * (32 * MT * 24);
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, 32 * MT * 24 * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) a3[mt][pc] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + (mt * 3 + pc) * 1024, 0, 0);
  };
  u32x4 xq[2];
  int wofs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int w = ((wave >> 2) * 64 + lane) * 4 + e;
    const int wc = w < 5 * P ? w : 0;
    wofs[e] = (wc % 5) * PLANE + (wave & 3) * QROW + (wc / 5) * 4;
  }
  auto stage_load = [&](int chunk) {
    const float* xp = X + ((size_t)((blockIdx.x >> 3) * 64 + (chunk & 63)) * 16 + 4 * (wave & 3)) * 512 + ((wave >> 2) & 1) * 256 + lane * 4;
#pragma unroll
    for (int r = 0; r < 2; ++r) xq[r] = *reinterpret_cast<const u32x4*>(xp + (r + 2 * (wave >> 3)) * 512);
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      *reinterpret_cast<f32x2*>(lds + buf * SLAB + wofs[e] + 2 * (wave >> 3)) = (f32x2){__uint_as_float(xq[0][e]), __uint_as_float(xq[1][e])};
  };
  load_a(0);
  int xbuf = 0, grp = 0;
  for (int c = 0; c < chunks; ++c) {
    stage_load(c + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < GC; ++g) {
      float v[8];
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        f32x4 x[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int cc = 4 * g + j + 1;
          x[j] = *reinterpret_cast<const f32x4*>(lds + xbuf * SLAB + lane_base + hq * QROW + (cc % 5) * PLANE + (cc / 5) * 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = x[5][e];
#pragma unroll
          for (int j = 0; j < 5; ++j) t = __builtin_fmaf(co[j], x[j][e], t);
          v[4 * hq + e] = t;
        }
      }
      u32x4 pc[3];
      split8<false>(v, pc[0], pc[1], pc[2]);
      const bf16x8 bh = __builtin_bit_cast(bf16x8, pc[0]), bm = __builtin_bit_cast(bf16x8, pc[1]), bl = __builtin_bit_cast(bf16x8, pc[2]);
#pragma unroll
      for (int pp = 0; pp < 6; ++pp) {
        constexpr int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a3[mt][pa[pp]]), pb[pp] == 0 ? bh : pb[pp] == 1 ? bm : bl,
                                                            acc[mt], 0, 0, 0);
      }
      load_a(grp + 1);
      ++grp;
      __builtin_amdgcn_sched_barrier(0);
    }
    stage_store(xbuf ^ 1);
    __syncthreads();
    xbuf ^= 1;
  }
  float s_ = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_ += acc[i][r];
  out[(size_t)blockIdx.x * 1024 + tid] = s_;
}

template <int MT, int GC>
static void run16(int chunks) {
  const int blocks = 256;
  const int panel = (chunks + 2) * GC * 8 * 32 * MT * 24;
  CHECK(hipFuncSetAttribute((const void*)loop54w16<MT, GC>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLAB * 4 + 64));
  float* out16;
  CHECK(hipMalloc(&out16, (size_t)blocks * 1024 * 4));
  const double ms = time_ms([&] {
    hipLaunchKernelGGL((loop54w16<MT, GC>), dim3(blocks), dim3(1024), 2 * SLAB * 4 + 64, 0, dA, dX, out16, chunks, panel);
  }, 150);
  CHECK(hipGetLastError());
  const double mfma = (double)chunks * GC * 6 * MT;                       // per wave (one column)
  const double tf = 2.0 * 32 * 32 * 16 * mfma / 6.0 * 16 * blocks / (ms * 1e-3) / 1e12;
  printf("MT %d GC %d 16 waves = (point, column), 4 per SIMD    %8.3f ms  %7.1f fp32-eq TFLOP/s  (%.3f us per tap group of %d rows x 2 columns)\n",
         MT, GC, ms, tf, ms * 1e3 / (chunks * GC), 32 * MT);
  CHECK(hipFree(out16));
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < reps / 2 + 1; ++i) launch();          // warm-up: the clock needs ~20 ms of load after idle
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int MT, int MASK, int GC>
static void run(const char* what, int chunks) {
  const int blocks = 256;
  const int panel = (chunks + 2) * GC * 8 * 32 * MT * 24;      // floats per weight panel
  CHECK(hipFuncSetAttribute((const void*)loop54<MT, MASK, GC>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLAB * 4 + 64));
  const double ms = time_ms([&] {
    hipLaunchKernelGGL((loop54<MT, MASK, GC>), dim3(blocks), dim3(THREADS), 2 * SLAB * 4 + 64, 0, dA, dX, dOut, dClk, chunks, panel);
  }, 150);
  CHECK(hipGetLastError());
  std::vector<unsigned long long> clk(blocks);
  CHECK(hipMemcpy(clk.data(), dClk, blocks * 8, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (auto c : clk) cyc += (double)c;
  cyc /= blocks;
  std::vector<unsigned long long> rt(blocks);
  CHECK(hipMemcpy(rt.data(), dClk + 256 + 256 * 6, blocks * 8, hipMemcpyDeviceToHost));
  double rtt = 0;
  for (auto c : rt) rtt += (double)c;
  rtt /= blocks;
  const double ghz = cyc / (rtt / 100e6) / 1e9;            // s_memtime ticks per second of s_memrealtime (100 MHz)
  const double mfma = (double)chunks * GC * 2 * 6 * MT;                    // per wave
  const double floor_cyc = mfma * 32 * 2;                                   // 2 waves per SIMD share the pipe
  const double tf = 2.0 * 32 * 32 * 16 * mfma / 6.0 * 8 * blocks / (ms * 1e-3) / 1e12;     // fp32-equivalent (6 MFMAs = 1 product)
  // (s_memtime ticks: the MFMA-only loop measures 16.3 ticks per MFMA and wave with two waves per SIMD, i.e. one tick = two
  // cycles of the 32-cycle instruction: busy = 32 ticks per item-pair MFMA)
  if (MASK & 32768) {
    std::vector<unsigned long long> tr(256 * 6);
    CHECK(hipMemcpy(tr.data(), dClk + 256, 256 * 6 * 8, hipMemcpyDeviceToHost));
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < 256; ++b) for (int k = 0; k < 6; ++k) a[k] += (double)tr[b * 6 + k] / 256;
    const double items = (double)chunks * GC * 2;
    printf("      trace, ticks per item: wave 0 prep %.0f mul %.0f sync %.0f | wave 7 prep %.0f mul %.0f sync %.0f\n", a[0] / items, a[1] / items,
           a[2] / items, a[3] / items, a[4] / items, a[5] / items);
  }
  printf("MT %d GC %d %-30s %8.3f ms  %7.1f fp32-eq TFLOP/s incl. launch  | loop: %5.0f ticks per item, %.2f G ticks/s, loop %.3f ms\n", MT, GC,
         what, ms, tf, cyc / (chunks * GC * 2), ghz, rtt / 100e3);
}

int main(int argc, char** argv) {
  const int chunks = argc > 1 ? atoi(argv[1]) : 48;
  const size_t a_floats = (size_t)8 * (chunks + 2) * 3 * 8 * 32 * 4 * 24 + 4096;
  CHECK(hipMalloc(&dA, a_floats * 4));
  CHECK(hipMalloc(&dX, (size_t)256 * 64 * 16 * 512 * 4 + 65536));
  CHECK(hipMalloc(&dOut, (size_t)256 * THREADS * 4));
  CHECK(hipMalloc(&dClk, (256 + 256 * 6 + 256) * 8));
  {
    std::vector<unsigned> h(a_floats);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c003b80u + (unsigned)((i * 2654435761u) >> 20 & 0x00ff00ffu);
    CHECK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hx((size_t)256 * 64 * 16 * 512 + 16384);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0.01f * (float)((i * 31) % 97) - 0.4f;
    CHECK(hipMemcpy(dX, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  }
#define RUNS(MT, GC)                                                            \
  run<MT, 16384 | 65536, GC>("product form (A behind, PIN ALAY)", chunks);      \
  run<MT, 16384, GC>("product form, row-major weights", chunks);                \
  run<MT, 1 | 4 | 8 | 16 | 32 | 64, GC>("MFMA only", chunks);                   \
  run<MT, 1 | 128 | 16384, GC>("everything but the MFMAs", chunks);
  if (argc > 2 && argv[2][0] == 'x') {        // where the non-MFMA work's time goes: one ingredient out at a time, with the phase trace
    run<3, 16384 | 65536 | 32768, 2>("product form + trace", chunks);
    run<3, 16384 | 65536 | 32768 | 1024, 2>("... no barrier (racy) + trace", chunks);
    run<3, 1 | 128 | 16384, 2>("no MFMA", chunks);
    run<3, 1 | 128 | 16384 | 32768, 2>("no MFMA + trace", chunks);
    run<3, 1 | 128 | 16384 | 4, 2>("no MFMA, no weight loads", chunks);
    run<3, 1 | 128 | 16384 | 64, 2>("no MFMA, no staging / barrier", chunks);
    run<3, 1 | 128 | 16384 | 4 | 64, 2>("no MFMA, no weights, no staging", chunks);
    run<3, 1 | 128 | 16384 | 4 | 64 | 32768, 2>("no MFMA, no weights, no staging + trace", chunks);
    run<3, 1 | 128 | 16384 | 4 | 64 | 32, 2>("no MFMA / weights / staging / split", chunks);
    run<3, 1 | 128 | 16384 | 4 | 64 | 16, 2>("no MFMA / weights / staging / transform", chunks);
    run<3, 1 | 128 | 16384 | 4 | 64 | 16 | 32, 2>("LDS reads only", chunks);
    return 0;
  }
  RUNS(3, 2)
  RUNS(2, 2)
  run16<2, 2>(chunks);
  run16<3, 2>(chunks);
  run16<2, 1>(chunks);
  run16<2, 3>(chunks);
  return 0;
}
