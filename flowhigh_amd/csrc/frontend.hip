// HBM-bound kernels either side of the DFT GEMMs: STFT framing, spectral energy / cutoff search,
// band splice, inverse-STFT overlap-add, peak normalisation, polyphase resampling.
//
// Replaces, in the reference (paths under /root/reference/src/flowhigh/):
//   fh_frame_f32          F.pad(reflect) + the framing half of torch.stft   models/melvoco.py:74-79
//                         and of torchaudio Spectrogram (zero pad)           postprocessing.py:7,22-23
//   fh_spec_energy_f32 /  get_cutoff_index (a <=1025-iteration python loop
//   fh_cutoff_index_f32   with a host sync per iteration on GPU)             postprocessing.py:10-16
//   fh_spec_splice_f32    result[:cr] = src ; result[cr:] = pred            postprocessing.py:36-37
//   fh_istft_ola_f32      the overlap-add half of torch.istft               postprocessing.py:8,39
//   fh_peak_*             audio / max|audio| * 0.99 ; cond /= max|cond|      postprocessing.py:40, flowhighsr.py:69
//   fh_resample_poly_f32  scipy.signal.resample_poly (host numpy in the ref) flowhighsr.py:68
#include "fh_common.h"

namespace {

constexpr int P_BLOCKS = 33;               // 33 * 32 = 1056 >= 1025 bins
constexpr int P_WIDTH = P_BLOCKS * 64;     // 2112 floats per frame

__global__ __launch_bounds__(256) void frame_kernel(const float* __restrict__ audio,
                                                    const float* __restrict__ window,
                                                    float* __restrict__ frames, int len,
                                                    int n_frames, int nfft, int hop, int pad,
                                                    int pad_mode) {
  const int t = blockIdx.x, b = blockIdx.y;
  const float* a = audio + (size_t)b * len;
  float* f = frames + ((size_t)b * n_frames + t) * nfft;
  for (int k = threadIdx.x; k < nfft; k += 256) {
    int i = hop * t + k - pad;
    float v;
    if (pad_mode == 0) {            // reflect (no edge repeat); pad < len
      if (i < 0) i = -i;
      if (i >= len) i = 2 * (len - 1) - i;
      v = a[i];
    } else {
      v = (i >= 0 && i < len) ? a[i] : 0.f;
    }
    f[k] = v * window[k];
  }
}

// grid (33, batch); thread -> (bin i = tid & 31, frame lane = tid >> 5); 32 frame lanes x 2 independent
// partial sums keep enough loads in flight (8 lanes with one dependent chain each took 46 us at B = 1)
constexpr int SE_LANES = 32;
__global__ __launch_bounds__(32 * SE_LANES) void spec_energy_kernel(const float* __restrict__ spec,
                                                                    float* __restrict__ energy,
                                                                    int n_frames) {
  __shared__ double part[SE_LANES][32];
  const int blk = blockIdx.x, b = blockIdx.y;
  const int i = threadIdx.x & 31, fl = threadIdx.x >> 5;
  const float* s = spec + (size_t)b * n_frames * P_WIDTH + blk * 64;
  double acc = 0.0, acc2 = 0.0;
  for (int t = fl; t < n_frames; t += 2 * SE_LANES) {
    float re = s[(size_t)t * P_WIDTH + i], im = s[(size_t)t * P_WIDTH + 32 + i];
    acc += (double)sqrtf(re * re + im * im);
    const int t2 = t + SE_LANES;
    if (t2 < n_frames) {
      re = s[(size_t)t2 * P_WIDTH + i], im = s[(size_t)t2 * P_WIDTH + 32 + i];
      acc2 += (double)sqrtf(re * re + im * im);
    }
  }
  part[fl][i] = acc + acc2;
  __syncthreads();
  if (fl == 0) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < SE_LANES; ++q) tot += part[q][i];
    int bin = blk * 32 + i;
    if (bin < 1025) energy[b * 1025 + bin] = (float)tot;
  }
}

// torch.cumsum on CPU accumulates float32 input in double and rounds every prefix to float;
// the threshold product is a float32 multiply (postprocessing.py:11-12).
__global__ __launch_bounds__(64) void cutoff_kernel(const float* __restrict__ energy, int32_t* __restrict__ cr,
                                                    int nbins, float thr) {
  // One wave per clip: lane l owns the contiguous chunk [l * per, (l + 1) * per) of the bins; chunk
  // sums are scanned across lanes in double, then every prefix is rounded to float exactly where the
  // sequential double accumulation of torch.cumsum would round it (same values up to 1e-16 relative).
  __shared__ float cum[1088];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int per = (nbins + 63) / 64;
  const int f0 = lane * per;
  double local = 0.0;
  for (int f = f0; f < f0 + per && f < nbins; ++f) local += (double)energy[b * nbins + f];
  double incl = local;                              // inclusive scan of the chunk sums
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  double c = incl - local;
  for (int f = f0; f < f0 + per && f < nbins; ++f) {
    c += (double)energy[b * nbins + f];
    cum[f] = (float)c;
  }
  __syncthreads();
  const float limit = cum[nbins - 1] * thr;
  // largest j in [1, nbins - 1] with cum[j] < limit, else 0
  int best = 0;
  for (int f = f0; f < f0 + per && f < nbins; ++f)
    if (f >= 1 && cum[f] < limit) best = f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(best, o, 64);
    best = other > best ? other : best;
  }
  if (lane == 0) cr[b] = best;
}

// energy[b, d] = sum_n exp(mel[b, n, d])   (locate_cutoff_freq on exp(mel), cfm_superresolution.py:134-159)
// seg != nullptr: clips of different lengths packed back to back, clip b = rows seg[2b] .. seg[2b] + seg[2b+1]
// (the ragged serving path); the sum runs over the clip's own rows in the same order either way.
__global__ __launch_bounds__(256) void mel_energy_kernel(const float* __restrict__ mel,
                                                         float* __restrict__ energy, int n, int d,
                                                         const int32_t* __restrict__ seg) {
  const int b = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= d) return;
  const size_t row0 = seg ? (size_t)seg[2 * b] : (size_t)b * n;
  const int rows = seg ? seg[2 * b + 1] : n;
  const float* m = mel + row0 * d + col;
  double acc = 0.0;
  for (int t = 0; t < rows; ++t) acc += (double)expf(m[(size_t)t * d]);
  energy[b * d + col] = (float)acc;
}

// out[b, n, d] = d < cut[b] ? low : high      (mel_replace_ops, cfm_superresolution.py:146-152)
__global__ __launch_bounds__(256) void mel_splice_kernel(const float* __restrict__ low,
                                                         const float* __restrict__ high,
                                                         const int32_t* __restrict__ cut,
                                                         float* __restrict__ out, int n, int d,
                                                         const int32_t* __restrict__ seg) {
  const int b = blockIdx.y;
  const size_t per = (size_t)(seg ? seg[2 * b + 1] : n) * d;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  const size_t g = (seg ? (size_t)seg[2 * b] : (size_t)b * n) * d + i;
  out[g] = (int)(i % d) < cut[b] ? low[g] : high[g];
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float a,
                                                    const float* __restrict__ y, float bcoef,
                                                    float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = x[i] * a + y[i] * bcoef;      // cond * std_1 + epsilon * std_2 (cfm:226-236)
}

__global__ __launch_bounds__(256) void splice_kernel(const float* __restrict__ pred,
                                                     const float* __restrict__ src,
                                                     const int32_t* __restrict__ cr,
                                                     float* __restrict__ out, int n_frames) {
  const int b = blockIdx.y;
  const size_t per_clip = (size_t)n_frames * P_WIDTH;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= per_clip) return;
  const int col = (int)(idx % P_WIDTH);
  const int bin = (col >> 6) * 32 + (col & 31);
  const size_t g = (size_t)b * per_clip + idx;
  out[g] = bin < cr[b] ? src[g] : pred[g];
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames,
                                                        const float* __restrict__ window,
                                                        float* __restrict__ y,
                                                        uint32_t* __restrict__ peak_bits,
                                                        int n_frames, int len, int nfft, int hop) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (j < len) {
    const int avail = hop * (n_frames - 1) + nfft / 2;   // torch.istft zero-fills past the OLA signal's end
    if (j < avail) {
      const int p = j + nfft / 2;
      int t_hi = p / hop;
      if (t_hi > n_frames - 1) t_hi = n_frames - 1;
      int t_lo = (p - nfft + hop) / hop;          // ceil((p - nfft + 1) / hop) for p - nfft + 1 > 0
      if (p - nfft + 1 <= 0) t_lo = 0;
      float num = 0.f, den = 0.f;
      const float* fb = frames + (size_t)b * n_frames * nfft;
      for (int t = t_lo; t <= t_hi; ++t) {
        const int k = p - hop * t;
        const float w = window[k];
        num = fmaf(w, fb[(size_t)t * nfft + k], num);
        den = fmaf(w, w, den);
      }
      v = num / den;
    }
    y[(size_t)b * len + j] = v;
  }
  float m = wave_max(fabsf(v));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(peak_bits + b, __float_as_uint(m));
  }
}

__global__ __launch_bounds__(256) void peak_abs_kernel(const float* __restrict__ x,
                                                       uint32_t* __restrict__ peak_bits, int len) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const float* xb = x + (size_t)b * len;
  float m = 0.f;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < len; j += gridDim.x * 256) m = fmaxf(m, fabsf(xb[j]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(peak_bits + b, __float_as_uint(m));
  }
}

__global__ __launch_bounds__(256) void peak_scale_kernel(float* __restrict__ y,
                                                         const uint32_t* __restrict__ peak_bits,
                                                         int len, float target) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= len) return;
  const float peak = __uint_as_float(peak_bits[b]);
  float v = y[(size_t)b * len + j] / peak;       // same order as the reference: (y / peak) * 0.99
  y[(size_t)b * len + j] = v * target;
}

// out[i] = sum_j x[j] * h[(i + pre) * down - j * up],  h zero outside [0, n_taps)
__global__ __launch_bounds__(256) void resample_poly_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ h,
                                                            float* __restrict__ y, int len_in,
                                                            int len_out, int up, int down,
                                                            int n_taps, int pre) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= len_out) return;
  const float* xb = x + (size_t)b * len_in;
  const long long pos = (long long)(i + pre) * down;
  long long j_hi = pos / up;
  if (j_hi > len_in - 1) j_hi = len_in - 1;
  float acc = 0.f;
  // ascending j == descending tap index; scipy's upfirdn walks the taps in ascending order,
  // so accumulate from the smallest tap index (largest j) down to match its summation order.
  for (long long j = j_hi; j >= 0; --j) {
    long long k = pos - j * up;
    if (k >= n_taps) break;
    acc = fmaf(xb[j], h[k], acc);
  }
  y[(size_t)b * len_out + i] = acc;
}

}  // namespace

extern "C" int fh_frame_f32(const float* audio, const float* window, float* frames, int batch,
                            int len, int n_frames, int nfft, int hop, int pad, int pad_mode,
                            void* stream) {
  FH_CHECK_ARG(audio && window && frames && batch > 0 && len > 0 && n_frames > 0, "fh_frame_f32: bad args");
  FH_CHECK_ARG(pad_mode == 1 || pad < len, "fh_frame_f32: reflect pad %d needs len > pad", pad);
  FH_CHECK_ARG(hop * (n_frames - 1) + nfft <= len + 2 * pad, "fh_frame_f32: frames exceed padded signal");
  hipLaunchKernelGGL(frame_kernel, dim3(n_frames, batch), dim3(256), 0, (hipStream_t)stream, audio,
                     window, frames, len, n_frames, nfft, hop, pad, pad_mode);
  FH_CHECK_LAUNCH("fh_frame_f32");
  return FH_OK;
}

extern "C" int fh_spec_energy_f32(const float* spec, float* energy, int batch, int n_frames,
                                  void* stream) {
  FH_CHECK_ARG(spec && energy && batch > 0 && n_frames > 0, "fh_spec_energy_f32: bad args");
  hipLaunchKernelGGL(spec_energy_kernel, dim3(P_BLOCKS, batch), dim3(32 * SE_LANES), 0, (hipStream_t)stream, spec,
                     energy, n_frames);
  FH_CHECK_LAUNCH("fh_spec_energy_f32");
  return FH_OK;
}

extern "C" int fh_cutoff_index_f32(const float* energy, int32_t* cr, int batch, int nbins, float thr,
                                   void* stream) {
  FH_CHECK_ARG(energy && cr && batch > 0 && nbins > 1 && nbins <= 1025, "fh_cutoff_index_f32: bad args");
  hipLaunchKernelGGL(cutoff_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, energy, cr, nbins, thr);
  FH_CHECK_LAUNCH("fh_cutoff_index_f32");
  return FH_OK;
}

extern "C" int fh_mel_energy_f32(const float* mel, float* energy, int batch, int n, int d, void* stream) {
  FH_CHECK_ARG(mel && energy && batch > 0 && n > 0 && d > 0, "fh_mel_energy_f32: bad args");
  hipLaunchKernelGGL(mel_energy_kernel, dim3(fh_cdiv(d, 256), batch), dim3(256), 0, (hipStream_t)stream, mel,
                     energy, n, d, (const int32_t*)nullptr);
  FH_CHECK_LAUNCH("fh_mel_energy_f32");
  return FH_OK;
}

extern "C" int fh_mel_energy_seg_f32(const float* mel, float* energy, const int32_t* seg, int n_seg, int d,
                                     void* stream) {
  FH_CHECK_ARG(mel && energy && seg && n_seg > 0 && n_seg < 65536 && d > 0, "fh_mel_energy_seg_f32: bad args");
  hipLaunchKernelGGL(mel_energy_kernel, dim3(fh_cdiv(d, 256), n_seg), dim3(256), 0, (hipStream_t)stream, mel,
                     energy, 0, d, seg);
  FH_CHECK_LAUNCH("fh_mel_energy_seg_f32");
  return FH_OK;
}

extern "C" int fh_mel_splice_f32(const float* low, const float* high, const int32_t* cut, float* out,
                                 int batch, int n, int d, void* stream) {
  FH_CHECK_ARG(low && high && cut && out && batch > 0 && n > 0 && d > 0, "fh_mel_splice_f32: bad args");
  hipLaunchKernelGGL(mel_splice_kernel, dim3(fh_cdiv((long long)n * d, 256), batch), dim3(256), 0,
                     (hipStream_t)stream, low, high, cut, out, n, d, (const int32_t*)nullptr);
  FH_CHECK_LAUNCH("fh_mel_splice_f32");
  return FH_OK;
}

extern "C" int fh_mel_splice_seg_f32(const float* low, const float* high, const int32_t* cut, float* out,
                                     const int32_t* seg, int n_seg, int max_n, int d, void* stream) {
  FH_CHECK_ARG(low && high && cut && out && seg && n_seg > 0 && n_seg < 65536 && max_n > 0 && d > 0,
               "fh_mel_splice_seg_f32: bad args");
  hipLaunchKernelGGL(mel_splice_kernel, dim3(fh_cdiv((long long)max_n * d, 256), n_seg), dim3(256), 0,
                     (hipStream_t)stream, low, high, cut, out, 0, d, seg);
  FH_CHECK_LAUNCH("fh_mel_splice_seg_f32");
  return FH_OK;
}

extern "C" int fh_axpby_f32(const float* x, float a, const float* y, float b, float* out, long long n,
                            void* stream) {
  FH_CHECK_ARG(x && y && out && n > 0, "fh_axpby_f32: bad args");
  hipLaunchKernelGGL(axpby_kernel, dim3(fh_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, a, y, b, out, n);
  FH_CHECK_LAUNCH("fh_axpby_f32");
  return FH_OK;
}

extern "C" int fh_spec_splice_f32(const float* pred, const float* src, const int32_t* cr, float* out,
                                  int batch, int n_frames, void* stream) {
  FH_CHECK_ARG(pred && src && cr && out && batch > 0 && n_frames > 0, "fh_spec_splice_f32: bad args");
  dim3 grid(fh_cdiv((long long)n_frames * P_WIDTH, 256), batch);
  hipLaunchKernelGGL(splice_kernel, grid, dim3(256), 0, (hipStream_t)stream, pred, src, cr, out, n_frames);
  FH_CHECK_LAUNCH("fh_spec_splice_f32");
  return FH_OK;
}

extern "C" int fh_istft_ola_f32(const float* frames, const float* window, float* y,
                                uint32_t* peak_bits, int batch, int n_frames, int len, int nfft,
                                int hop, void* stream) {
  FH_CHECK_ARG(frames && window && y && peak_bits && batch > 0 && n_frames > 0 && len > 0, "fh_istft_ola_f32: bad args");
  dim3 grid(fh_cdiv(len, 256), batch);
  hipLaunchKernelGGL(istft_ola_kernel, grid, dim3(256), 0, (hipStream_t)stream, frames, window, y,
                     peak_bits, n_frames, len, nfft, hop);
  FH_CHECK_LAUNCH("fh_istft_ola_f32");
  return FH_OK;
}

extern "C" int fh_peak_abs_f32(const float* x, uint32_t* peak_bits, int batch, int len, void* stream) {
  FH_CHECK_ARG(x && peak_bits && batch > 0 && len > 0, "fh_peak_abs_f32: bad args");
  int bx = fh_cdiv(len, 256);
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(peak_abs_kernel, dim3(bx, batch), dim3(256), 0, (hipStream_t)stream, x, peak_bits, len);
  FH_CHECK_LAUNCH("fh_peak_abs_f32");
  return FH_OK;
}

extern "C" int fh_peak_scale_f32(float* y, const uint32_t* peak_bits, int batch, int len,
                                 float target, void* stream) {
  FH_CHECK_ARG(y && peak_bits && batch > 0 && len > 0, "fh_peak_scale_f32: bad args");
  hipLaunchKernelGGL(peak_scale_kernel, dim3(fh_cdiv(len, 256), batch), dim3(256), 0, (hipStream_t)stream,
                     y, peak_bits, len, target);
  FH_CHECK_LAUNCH("fh_peak_scale_f32");
  return FH_OK;
}

extern "C" int fh_resample_poly_f32(const float* x, const float* taps, float* y, int batch,
                                    int len_in, int len_out, int up, int down, int n_taps,
                                    int n_pre_remove, void* stream) {
  FH_CHECK_ARG(x && taps && y && batch > 0 && len_in > 0 && len_out > 0 && up > 0 && down > 0, "fh_resample_poly_f32: bad args");
  hipLaunchKernelGGL(resample_poly_kernel, dim3(fh_cdiv(len_out, 256), batch), dim3(256), 0,
                     (hipStream_t)stream, x, taps, y, len_in, len_out, up, down, n_taps, n_pre_remove);
  FH_CHECK_LAUNCH("fh_resample_poly_f32");
  return FH_OK;
}
