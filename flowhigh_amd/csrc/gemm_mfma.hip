// fp32 GEMM on the gfx950 matrix cores:  C[M, N] = epilogue(A[M, K] * W[N, K]^T).
//
// Replaces the nn.Linear call sites of the FLowHigh transformer
// (/root/reference/src/flowhigh/models/flow.py:239,261; attend.py:170-171,176,189;
// transformer.py:98-104), and, used as DFT-by-GEMM, torch.stft / torch.istft
// (melvoco.py:78-79; postprocessing.py:22-23,39) and the mel projection (melvoco.py:83).
//
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  Block = 4 waves (2 x 2), BK = 32.
// Both operands are K-contiguous row tiles; rows are padded to 36 floats in LDS so that one
// ds_read_b128 per lane is conflict free (start bank 4 (9 i mod 16)) and feeds four k-steps; the K
// order inside a 32-wide tile is permuted identically for A and W (k-step (q, e) pairs columns
// 8q + e and 8q + 4 + e).  Global loads of tile i+1 are issued into registers before the MFMAs of
// tile i (register prefetch), LDS is single buffered, two barriers per tile.
//
// Tile variants: 128 x 128 (wave 64 x 64), 64 x 128 (wave 32 x 64) and 64 x 64 (wave 32 x 32) -- the
// smaller ones keep more CUs busy at M = N_frames ~ 1000.  W is padded to a multiple of 128 rows.
#include "fh_common.h"

namespace {

constexpr int BK = 32;
constexpr int LP = BK + 4;   // LDS row pitch, floats

__device__ __forceinline__ float epi_pair(float first, float second, int mode) {
  if (mode == FH_EPI_GEGLU) return gelu_erf(second) * first;
  return sqrtf(first * first + second * second + 1e-9f);
}

template <int MT, int NT>   // wave tile = (32 MT) x (32 NT), block tile = (64 MT) x (64 NT)
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, int lda,
                                                   const float* __restrict__ W,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ R, int ldr,
                                                   float* __restrict__ C, int ldc, int M, int N,
                                                   int K, float alpha, int mode, int m_tiles) {
  constexpr int BM = 64 * MT, BN = 64 * NT;
  constexpr int AREG = BM * 8 / 256;   // float4 per thread for the A tile
  constexpr int WREGS = BN * 8 / 256;  // float4 per thread for the W tile
  __shared__ __attribute__((aligned(16))) float As[BM * LP];
  __shared__ __attribute__((aligned(16))) float Ws[BN * LP];

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

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[AREG], wreg[WREGS];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AREG; ++i) {
      int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m0 + row < M) v = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + row) * lda + k0 + 4 * c4);
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WREGS; ++i) {
      int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
      wreg[i] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + row) * K + k0 + 4 * c4);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < AREG; ++i) {
      int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
      *reinterpret_cast<f32x4*>(As + row * LP + 4 * c4) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < WREGS; ++i) {
      int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
      *reinterpret_cast<f32x4*>(Ws + row * LP + 4 * c4) = wreg[i];
    }
  };

  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();          // previous tile's fragment reads are done
    lstore();
    __syncthreads();
    if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a[MT], bfr[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        a[mt] = *reinterpret_cast<const f32x4*>(As + ((wm * MT + mt) * 32 + l31) * LP + 4 * (2 * q + lh));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bfr[nt] = *reinterpret_cast<const f32x4*>(Ws + ((wn * NT + nt) * 32 + l31) * LP + 4 * (2 * q + lh));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], bfr[nt][e], acc[mt][nt], 0, 0, 0);
    }
  }

  // ---- epilogue.  D reg r of lane l: row = (r&3) + 8 (r>>2) + 4 lh, col = l31 --------------
  // The MFMA computed D[i][j] = sum_k A[i][k] W[j][k] with i = A row (m), j = W row (n).
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
          C[(size_t)m * ldc + n_out] = epi_pair(acc[mt][0][r] + b1, acc[mt][NT - 1][r] + b2, mode);
        }
    }
  }
}

// ---- y = act(W x + b), one wave per output row ------------------------------------------
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ W,
                                                   const float* __restrict__ x,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ y, int N, int K, int act) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  const float* w = W + (size_t)row * K;
  float s = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
    f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
    s = fmaf(wv[0], xv[0], s);
    s = fmaf(wv[1], xv[1], s);
    s = fmaf(wv[2], xv[2], s);
    s = fmaf(wv[3], xv[3], s);
  }
  s = wave_sum(s);
  if (lane == 0) {
    s += bias ? bias[row] : 0.f;
    if (act == 1) s = s / (1.f + expf(-s));
    y[row] = s;
  }
}

__global__ void time_fourier_kernel(const float* __restrict__ w, float t, float* __restrict__ out,
                                    int half) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  // pos_emb.py:24: freqs = x * weights * 2 * math.pi, evaluated left to right in fp32
  float f = t * w[i];
  f = f * 2.0f;
  f = f * 3.14159265358979323846f;
  out[i] = sinf(f);
  out[half + i] = cosf(f);
}

}  // namespace

extern "C" int fh_gemm_f32(const float* A, int lda, const float* W, const float* bias,
                           const float* R, int ldr, float* C, int ldc, int M, int N, int K,
                           float alpha, int epilogue, void* stream) {
  FH_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0, "fh_gemm_f32: bad args");
  FH_CHECK_ARG(K % BK == 0, "fh_gemm_f32: K=%d must be a multiple of %d", K, BK);
  FH_CHECK_ARG(lda % 4 == 0 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0,
               "fh_gemm_f32: A/W must be 16-byte aligned with lda %% 4 == 0");
  FH_CHECK_ARG(epilogue >= 0 && epilogue <= 3, "fh_gemm_f32: unknown epilogue %d", epilogue);
  if (epilogue == FH_EPI_GEGLU || epilogue == FH_EPI_MAG)
    FH_CHECK_ARG(N % 64 == 0, "fh_gemm_f32: pair epilogue needs N %% 64 == 0");
  hipStream_t st = (hipStream_t)stream;
  const bool plain = epilogue == FH_EPI_LINEAR || epilogue == FH_EPI_LOGCLAMP;
  const long long t128 = (long long)fh_cdiv(M, 128) * fh_cdiv(N, 128);
  const long long t64x128 = (long long)fh_cdiv(M, 64) * fh_cdiv(N, 128);
  if (plain && t64x128 < 200) {
    // few tiles (M = frames ~ 1000, N = 1024): 64 x 64 tiles put a block on every CU
    const int m_tiles = fh_cdiv(M, 64);
    const int blocks = fh_cdiv((long long)m_tiles * fh_cdiv(N, 64), 8) * 8;
    hipLaunchKernelGGL((gemm_kernel<1, 1>), dim3(blocks), dim3(256), 0, st, A, lda, W, bias, R, ldr, C, ldc,
                       M, N, K, alpha, epilogue, m_tiles);
  } else if (t128 < 512) {
    const int m_tiles = fh_cdiv(M, 64);
    const int blocks = fh_cdiv((long long)m_tiles * fh_cdiv(N, 128), 8) * 8;
    hipLaunchKernelGGL((gemm_kernel<1, 2>), dim3(blocks), dim3(256), 0, st, A, lda, W, bias, R, ldr, C, ldc,
                       M, N, K, alpha, epilogue, m_tiles);
  } else {
    const int m_tiles = fh_cdiv(M, 128);
    const int blocks = fh_cdiv((long long)m_tiles * fh_cdiv(N, 128), 8) * 8;
    hipLaunchKernelGGL((gemm_kernel<2, 2>), dim3(blocks), dim3(256), 0, st, A, lda, W, bias, R, ldr, C, ldc,
                       M, N, K, alpha, epilogue, m_tiles);
  }
  FH_CHECK_LAUNCH("fh_gemm_f32");
  return FH_OK;
}

extern "C" int fh_gemv_f32(const float* W, const float* x, const float* bias, float* y, int N,
                           int K, int act, void* stream) {
  FH_CHECK_ARG(W && x && y && N > 0 && K > 0 && K % 4 == 0, "fh_gemv_f32: bad args");
  hipLaunchKernelGGL(gemv_kernel, dim3(fh_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, W, x, bias, y,
                     N, K, act);
  FH_CHECK_LAUNCH("fh_gemv_f32");
  return FH_OK;
}

extern "C" int fh_time_fourier_f32(const float* w, float t, float* out, int half_dim, void* stream) {
  FH_CHECK_ARG(w && out && half_dim > 0, "fh_time_fourier_f32: bad args");
  hipLaunchKernelGGL(time_fourier_kernel, dim3(fh_cdiv(half_dim, 256)), dim3(256), 0,
                     (hipStream_t)stream, w, t, out, half_dim);
  FH_CHECK_LAUNCH("fh_time_fourier_f32");
  return FH_OK;
}
