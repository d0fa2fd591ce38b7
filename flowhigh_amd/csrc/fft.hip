// 2048-point real FFT / inverse real FFT of framed audio, one frame per block, in LDS.
//
// Replaces the DFT halves of torch.stft / torch.istft as the reference calls them:
//   MelVoco.encode            /root/reference/src/flowhigh/models/melvoco.py:56-86   (n_fft 2048, |X| -> mel)
//   PostProcessing.stft/istft /root/reference/src/flowhigh/postprocessing.py:5-9,39  (complex spectra, C2R)
// (windowing / framing and the overlap-add stay in fh_frame_f32 / fh_istft_ola_f32).
//
// Radix-2 decimation in time on 2048 complex points held as two double arrays in LDS (32 KB): bit-reversed
// load, 11 butterfly stages, one barrier each, 4 butterflies per thread and stage.  Twiddles
// W^k = exp(-2 pi i k / 2048), k < 1024, come from a table computed in float64 on the host.
// 0.11 MFLOP per frame instead of the 8.4 MFLOP of a DFT-by-GEMM; the kernel is bound by its 8 KB read
// and 8 KB write per frame.  The butterflies run in float64: an fp32 FFT carries the rounding of the
// LARGEST bins into every output (absolute floor ~1e-6 of the spectral peak), and the log-mel of the
// empty band above the input's Nyquist is decided right there (torch's own fp32 FFT is 6e-4 off the
// float64 pipeline in those bins; tests/tools/mel_compare.py).  In float64 the result is the correctly rounded
// spectrum of the fp32 frames: 2.3e-4 from the float64 pipeline, 3.7e-4 from the reference.
//
// Spectrum layout ("P-layout", shared with fh_spec_energy_f32 / fh_spec_splice_f32): 33 blocks of 64
// floats = 32 Re then 32 Im of bins 32 b .. 32 b + 31; bins >= 1025 are zero.
#include "fh_common.h"

namespace {

constexpr int FFT_N = 2048;
constexpr int FFT_LOG = 11;
constexpr int FFT_BINS = 1025;
constexpr int FFT_P_WIDTH = 2112;     // 33 * 64
constexpr int FFT_MAG_WIDTH = 1056;   // 33 * 32

__device__ __forceinline__ int bitrev11(int n) { return (int)(__brev((unsigned)n) >> 21); }

// 11 in-place butterfly stages over re[], im[] (bit-reversed order in, natural order out)
__device__ __forceinline__ void fft2048_stages(double* re, double* im, const double2* __restrict__ tw, int tid) {
#pragma unroll 1
  for (int s = 1; s <= FFT_LOG; ++s) {
    const int half = 1 << (s - 1);
    const int tstep = FFT_N >> s;               // twiddle index stride: W_m^pos = W_2048^(pos * 2048 / m)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = tid + 256 * q;
      const int pos = j & (half - 1);
      const int i0 = ((j >> (s - 1)) << s) + pos, i1 = i0 + half;
      const double2 w = tw[pos * tstep];
      const double xr = re[i1], xi = im[i1];
      const double tr = w.x * xr - w.y * xi, ti = w.x * xi + w.y * xr;
      const double ar = re[i0], ai = im[i0];
      re[i0] = ar + tr;
      im[i0] = ai + ti;
      re[i1] = ar - tr;
      im[i1] = ai - ti;
    }
  }
  __syncthreads();
}

// mode 0: packed complex spectrum [rows, 2112]; mode 1: magnitudes [rows, 1056]
__global__ __launch_bounds__(256) void rfft2048_kernel(const float* __restrict__ frames,
                                                       const double2* __restrict__ tw,
                                                       float* __restrict__ out, int mode) {
  __shared__ double re[FFT_N];
  __shared__ double im[FFT_N];
  const int tid = threadIdx.x;
  const float* x = frames + (size_t)blockIdx.x * FFT_N;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int n = 4 * (tid + 256 * q);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = bitrev11(n + e);
      re[r] = (double)v[e];
      im[r] = 0.0;
    }
  }
  fft2048_stages(re, im, tw, tid);
  if (mode == 0) {
    float* o = out + (size_t)blockIdx.x * FFT_P_WIDTH;
    for (int c = tid; c < FFT_P_WIDTH; c += 256) {
      const int f = (c >> 6) * 32 + (c & 31);
      const double v = (c & 32) ? im[f < FFT_BINS ? f : 0] : re[f < FFT_BINS ? f : 0];
      o[c] = f < FFT_BINS ? (float)v : 0.f;
    }
  } else {
    float* o = out + (size_t)blockIdx.x * FFT_MAG_WIDTH;
    for (int f = tid; f < FFT_MAG_WIDTH; f += 256) {
      // sqrt(re^2 + im^2 + 1e-9) from the float32 spectrum, as melvoco.py:81 forms it (the 1e-9 is
      // comparable to |X|^2 in the empty band above the input's Nyquist)
      const float a = (float)re[f < FFT_BINS ? f : 0], b = (float)im[f < FFT_BINS ? f : 0];
      o[f] = f < FFT_BINS ? sqrtf(a * a + b * b + 1e-9f) : 0.f;
    }
  }
}

// C2R: x[n] = (1/N) sum_k X[k] e^{+2 pi i k n / N} with X[N - k] = conj X[k] and the imaginary parts of
// DC and Nyquist ignored (torch.istft / irfft) = Re FFT(conj X)[n] / N.
__global__ __launch_bounds__(256) void irfft2048_kernel(const float* __restrict__ spec,
                                                        const double2* __restrict__ tw,
                                                        float* __restrict__ frames) {
  __shared__ double re[FFT_N];
  __shared__ double im[FFT_N];
  const int tid = threadIdx.x;
  const float* s = spec + (size_t)blockIdx.x * FFT_P_WIDTH;
  for (int k = tid; k <= FFT_N / 2; k += 256) {
    const int c = (k >> 5) * 64 + (k & 31);
    const double a = s[c];
    const double b = (k == 0 || k == FFT_N / 2) ? 0.0 : (double)s[c + 32];
    const int r0 = bitrev11(k);
    re[r0] = a;
    im[r0] = -b;                       // conj X[k]
    if (k != 0 && k != FFT_N / 2) {
      const int r1 = bitrev11(FFT_N - k);
      re[r1] = a;                      // conj X[N - k] = X[k]
      im[r1] = b;
    }
  }
  fft2048_stages(re, im, tw, tid);
  float* x = frames + (size_t)blockIdx.x * FFT_N;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int n = 4 * (tid + 256 * q);
    const f32x4 v = {(float)(re[n] * (1.0 / FFT_N)), (float)(re[n + 1] * (1.0 / FFT_N)),
                     (float)(re[n + 2] * (1.0 / FFT_N)), (float)(re[n + 3] * (1.0 / FFT_N))};
    *reinterpret_cast<f32x4*>(x + n) = v;
  }
}

}  // namespace

extern "C" int fh_rfft2048_f32(const float* frames, const double* twiddles, float* out, int rows, int mode,
                               void* stream) {
  FH_CHECK_ARG(frames && twiddles && out && rows > 0 && (mode == 0 || mode == 1), "fh_rfft2048_f32: bad args");
  hipLaunchKernelGGL(rfft2048_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, frames,
                     (const double2*)twiddles, out, mode);
  FH_CHECK_LAUNCH("fh_rfft2048_f32");
  return FH_OK;
}

extern "C" int fh_irfft2048_f32(const float* spec, const double* twiddles, float* frames, int rows, void* stream) {
  FH_CHECK_ARG(spec && twiddles && frames && rows > 0, "fh_irfft2048_f32: bad args");
  hipLaunchKernelGGL(irfft2048_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, spec,
                     (const double2*)twiddles, frames);
  FH_CHECK_LAUNCH("fh_irfft2048_f32");
  return FH_OK;
}
