// GEMM in the bf16 x 6 form:  C[M, N] = epilogue(A[M, K] * W[N, K]^T) with fp32 in / out / accumulate and every product as six
// v_mfma_f32_32x32x16_bf16 over exact three-piece splits (x = h + m + l, h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); the pairs
// h h, h m, m h, h l, l h, m m; dropped terms <= 2^-24 |a b|): the transformer's linears of a conv_form = 'bf16x6' model.
//
// Replaces the same nn.Linear call sites as gemm_mfma.hip (/root/reference/src/flowhigh/models/flow.py:239,261; attend.py:170-171,
// 176,189; transformer.py:98-104) -- same epilogues (bias, alpha, residual; GEGLU pairs), same tile variants and block order.
//
// The recipe of narrow_bf.hip: an operand is split ONCE, on its way into LDS (A: by the block that stages the tile, 11 vector
// instructions per pair of values; W: on the host, packing.pack_gemm_bf_weight), and the K loop is ds_read_b128 + MFMA with no
// vector arithmetic.  LDS tile of an operand: 16-byte units (8 consecutive k as bf16) [piece][k-octet 0..3][row]: the fragment of a
// 32-row tile, one k-step (16) and one piece is 2 x 32 consecutive units -- conflict-free ds_read_b128.  BK = 32 = two k-steps;
// global loads of stage i + 1 are in registers before the MFMAs of stage i; LDS single-buffered, two barriers per stage.
// W in memory: [64-row granule][k stage][piece][k-octet][64 rows][8 bf16] (12 KB per granule and stage: a stage's W tile is a
// straight copy of 1-2 granules).
#include "fh_common.h"

namespace {

typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int GB_BK = 32;
constexpr int GB_GRAN = 3 * 4 * 64;        // 16-byte units of one (64-row granule, k stage) of packed W

__device__ __forceinline__ unsigned gb_pack(float a, float b) {
  const gb_bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float gb_lo(unsigned p) { return __uint_as_float(__builtin_amdgcn_perm(0u, p, 0x01000c0cu)); }
__device__ __forceinline__ float gb_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ void gb_split8(const f32x4& v0, const f32x4& v1, u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = i < 2 ? v0[2 * i] : v1[2 * i - 4], b = i < 2 ? v0[2 * i + 1] : v1[2 * i - 3];
    h[i] = gb_pack(a, b);
    const float ra = a - gb_lo(h[i]), rb = b - gb_hi(h[i]);
    m[i] = gb_pack(ra, rb);
    l[i] = gb_pack(ra - gb_lo(m[i]), rb - gb_hi(m[i]));
  }
}

__device__ __forceinline__ float gb_epi_pair(float first, float second, int mode) {
  if (mode == FH_EPI_GEGLU) return gelu_erf(second) * first;
  return sqrtf(first * first + second * second + 1e-9f);
}

template <int MT, int NT>   // wave tile = (32 MT) x (32 NT), block tile = (64 MT) x (64 NT)
__global__ __launch_bounds__(256) void gemm_bf_kernel(const float* __restrict__ A, int lda, const u32x4* __restrict__ Wp,
                                                      const float* __restrict__ bias, const float* __restrict__ R, int ldr,
                                                      float* __restrict__ C, int ldc, int M, int N, int K, float alpha, int mode,
                                                      int m_tiles) {
  constexpr int BM = 64 * MT, BN = 64 * NT;
  constexpr int AIT = MT;          // (row, k-octet) items of the A tile per thread: BM x 4 / 256
  constexpr int WIT = 3 * NT;      // 16-byte units of the W tile per thread: NT granules x 768 / 256
  constexpr int AP = BM + 4;       // pitch of a (piece, k-octet) plane of the A tile: a staging pass's 16 lanes are 4 rows x 4 k-octets,
                                   // 4 mod 16 puts them in 16 different bank groups (pitch BM: 4 to a group)
  __shared__ __attribute__((aligned(16))) u32x4 As[3 * 4 * AP];
  __shared__ __attribute__((aligned(16))) u32x4 Ws[3 * 4 * BN];

  // XCD-aware order: the m-tiles of one n-tile (sharing the W panel) go to one XCD
  const int bid = blockIdx.x;
  const int per_xcd = gridDim.x >> 3;              // grid is a multiple of 8
  const int work = (bid & 7) * per_xcd + (bid >> 3);
  const int nt_idx = work / m_tiles;
  const int mt_idx = work % m_tiles;
  const int m0 = mt_idx * BM, n0 = nt_idx * BN;
  if (n0 >= N) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int kstages = K / GB_BK;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[AIT][2];
  u32x4 wreg[WIT];
  auto gload = [&](int ks) {
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int item = tid + 256 * i, row = item >> 2, ko = item & 3;
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (m0 + row < M) {
        const float* p = A + (size_t)(m0 + row) * lda + ks * GB_BK + 8 * ko;
        v0 = *reinterpret_cast<const f32x4*>(p);
        v1 = *reinterpret_cast<const f32x4*>(p + 4);
      }
      areg[i][0] = v0;
      areg[i][1] = v1;
    }
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int u = tid + 256 * i, g = u / GB_GRAN, r = u - g * GB_GRAN;
      wreg[i] = Wp[((size_t)((n0 >> 6) + g) * kstages + ks) * GB_GRAN + r];
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int item = tid + 256 * i, row = item >> 2, ko = item & 3;
      u32x4 h, m, l;
      gb_split8(areg[i][0], areg[i][1], h, m, l);
      As[(0 * 4 + ko) * AP + row] = h;
      As[(1 * 4 + ko) * AP + row] = m;
      As[(2 * 4 + ko) * AP + row] = l;
    }
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int u = tid + 256 * i, g = u / GB_GRAN, r = u - g * GB_GRAN;
      Ws[(r >> 6) * BN + g * 64 + (r & 63)] = wreg[i];          // r >> 6 = piece * 4 + k-octet
    }
  };

  gload(0);
  for (int ks = 0; ks < kstages; ++ks) {
    __syncthreads();          // previous stage's fragment reads are done
    lstore();
    __syncthreads();
    if (ks + 1 < kstages) gload(ks + 1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      gb_bf16x8 a[MT][3], b[NT][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt][p] = __builtin_bit_cast(gb_bf16x8, As[(p * 4 + 2 * q + lh) * AP + (wm * MT + mt) * 32 + l31]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt][p] = __builtin_bit_cast(gb_bf16x8, Ws[(p * 4 + 2 * q + lh) * BN + (wn * NT + nt) * 32 + l31]);
      }
      // piece pairs (A piece, W piece), small terms first: (l h) (h l) (m m) (m h) (h m) (h h)
#pragma unroll
      for (int pp = 0; pp < 6; ++pp) {
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][pa[pp]], b[nt][pb[pp]], acc[mt][nt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue (as gemm_mfma.hip).  D reg r of lane l: row = (r&3) + 8 (r>>2) + 4 lh, col = l31 --------------------------
  if (mode == FH_EPI_LINEAR || mode == FH_EPI_LOGCLAMP) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 32 + l31;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m >= M) continue;
          float v = acc[mt][nt][r] + bv;
          if (mode == FH_EPI_LOGCLAMP) {
            v = logf(fmaxf(v, 1e-5f));
          } else {
            v *= alpha;
            if (R) v += R[(size_t)m * ldr + n];
          }
          C[(size_t)m * ldc + n] = v;
        }
      }
  } else {
    // pair modes: the wave's two 32-column tiles are (first, second) of one packed 64 block
    const int blk = (n0 >> 6) + wn;                 // packed block index
    const int n_out = blk * 32 + l31;
    const int n_first = n0 + wn * 64 + l31;         // packed column of `first`
    if (NT == 2 && n_first < N) {
      const float b1 = bias ? bias[n_first] : 0.f;
      const float b2 = bias ? bias[n_first + 32] : 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m >= M) continue;
          C[(size_t)m * ldc + n_out] = gb_epi_pair(acc[mt][0][r] + b1, acc[mt][NT - 1][r] + b2, mode);
        }
    }
  }
}

}  // namespace

extern "C" int fh_gemm_bf16x6_f32(const float* A, int lda, const float* Wp, const float* bias, const float* R, int ldr, float* C,
                                  int ldc, int M, int N, int K, float alpha, int epilogue, void* stream) {
  FH_CHECK_ARG(A && Wp && C && M > 0 && N > 0 && K > 0, "fh_gemm_bf16x6_f32: bad args");
  FH_CHECK_ARG(K % GB_BK == 0, "fh_gemm_bf16x6_f32: K=%d must be a multiple of %d", K, GB_BK);
  FH_CHECK_ARG(lda % 4 == 0 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)Wp) & 15) == 0,
               "fh_gemm_bf16x6_f32: A/W must be 16-byte aligned with lda %% 4 == 0");
  FH_CHECK_ARG(epilogue >= 0 && epilogue <= 3, "fh_gemm_bf16x6_f32: unknown epilogue %d", epilogue);
  if (epilogue == FH_EPI_GEGLU || epilogue == FH_EPI_MAG)
    FH_CHECK_ARG(N % 64 == 0, "fh_gemm_bf16x6_f32: pair epilogue needs N %% 64 == 0");
  hipStream_t st = (hipStream_t)stream;
  const u32x4* W4 = reinterpret_cast<const u32x4*>(Wp);
  const bool plain = epilogue == FH_EPI_LINEAR || epilogue == FH_EPI_LOGCLAMP;
  const long long t128 = (long long)fh_cdiv(M, 128) * fh_cdiv(N, 128);
  const long long t64x128 = (long long)fh_cdiv(M, 64) * fh_cdiv(N, 128);
  if (plain && t64x128 < 200) {
    const int m_tiles = fh_cdiv(M, 64);
    const int blocks = fh_cdiv((long long)m_tiles * fh_cdiv(N, 64), 8) * 8;
    hipLaunchKernelGGL((gemm_bf_kernel<1, 1>), dim3(blocks), dim3(256), 0, st, A, lda, W4, bias, R, ldr, C, ldc, M, N, K, alpha,
                       epilogue, m_tiles);
  } else if (t128 < 512) {
    const int m_tiles = fh_cdiv(M, 64);
    const int blocks = fh_cdiv((long long)m_tiles * fh_cdiv(N, 128), 8) * 8;
    hipLaunchKernelGGL((gemm_bf_kernel<1, 2>), dim3(blocks), dim3(256), 0, st, A, lda, W4, bias, R, ldr, C, ldc, M, N, K, alpha,
                       epilogue, m_tiles);
  } else {
    const int m_tiles = fh_cdiv(M, 128);
    const int blocks = fh_cdiv((long long)m_tiles * fh_cdiv(N, 128), 8) * 8;
    hipLaunchKernelGGL((gemm_bf_kernel<2, 2>), dim3(blocks), dim3(256), 0, st, A, lda, W4, bias, R, ldr, C, ldc, M, N, K, alpha,
                       epilogue, m_tiles);
  }
  FH_CHECK_LAUNCH("fh_gemm_bf16x6_f32");
  return FH_OK;
}
