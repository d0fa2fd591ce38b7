"""Log-mel front end, STFT-domain post-processing and the resampling pre-step on the HIP kernels.

  LogMel          MelVoco.encode                   /root/reference/src/flowhigh/models/melvoco.py:56-86
  PostProcessor   PostProcessing.post_processing   /root/reference/src/flowhigh/postprocessing.py:5-41
  Resampler       scipy.signal.resample_poly + peak normalise   flowhighsr.py:68-69

STFT / iSTFT are 2048-point FFTs in LDS (csrc/fft.hip; FH_FFT=0 selects the older DFT-by-GEMM on the
matrix cores, 8.4 MFLOP per frame) with the magnitude, the mel projection's log and the window /
overlap-add fused around them; the reference's python cutoff loop (one host sync per bin on GPU) is a
device scan.
"""
import os

import torch

from . import hip, tables
from .tables import HOP, MAG_WIDTH, N_FFT, N_MELS, P_WIDTH


_USE_FFT = os.environ.get("FH_FFT", "1") != "0"


class _Const:
    _cache = {}

    @classmethod
    def get(cls, device):
        device = hip.norm_device(device)
        key = (device.type, device.index)
        if key not in cls._cache:
            c = dict(hann=tables.hann_window().to(device), w_mel=tables.mel_gemm_weight().to(device))
            if _USE_FFT:
                c["tw"] = tables.fft_twiddles().to(device)
            else:               # FH_FFT=0: STFT / iSTFT as DFT-by-GEMM (35 MB of bases, not uploaded otherwise)
                c["w_fwd"] = tables.dft_forward_weight().to(device)
                c["w_inv"] = tables.dft_inverse_weight().to(device)
            cls._cache[key] = c
        return cls._cache[key]


class LogMel:
    def __init__(self, device):
        self.device = hip.norm_device(device)
        self.c = _Const.get(self.device)
        self._ws = hip.ShapeCache()

    @hip.on_device
    def __call__(self, audio):
        """audio [B, T] on device -> log-mel [B*N, 256] (token-major rows), N = T // 480."""
        B, T = audio.shape
        N = T // HOP
        if N < 1 or T <= (N_FFT - HOP) // 2:
            raise ValueError(f"clip of {T} samples is too short for the mel front end")
        key = (B, T)
        if key not in self._ws:
            f32 = dict(dtype=torch.float32, device=self.device)
            self._ws[key] = (torch.empty(B * N, N_FFT, **f32), torch.empty(B * N, MAG_WIDTH, **f32))
        frames, mag = self._ws[key]
        L, st = hip.lib(), hip.stream()
        audio = audio.contiguous()
        hip.check(L.fh_frame_f32(audio.data_ptr(), self.c["hann"].data_ptr(), frames.data_ptr(), B, T, N,
                                 N_FFT, HOP, (N_FFT - HOP) // 2, 0, st), "fh_frame_f32")
        if _USE_FFT:
            hip.check(L.fh_rfft2048_f32(frames.data_ptr(), self.c["tw"].data_ptr(), mag.data_ptr(), B * N, 1, st),
                      "fh_rfft2048_f32")
        else:
            hip.gemm(frames, self.c["w_fwd"], mag, B * N, P_WIDTH, N_FFT, epilogue=hip.EPI_MAG)
        mel = torch.empty(B * N, N_MELS, dtype=torch.float32, device=self.device)
        hip.gemm(mag, self.c["w_mel"], mel, B * N, N_MELS, MAG_WIDTH, epilogue=hip.EPI_LOGCLAMP)
        return mel


class PostProcessor:
    def __init__(self, device):
        self.device = hip.norm_device(device)
        self.c = _Const.get(self.device)
        self._ws = hip.ShapeCache()

    @hip.on_device
    def __call__(self, pred, src, length, return_cr=False):
        """pred [B, Tp], src [B, T] -> [B, length]; per-clip cutoff, splice, iSTFT, 0.99 peak."""
        B, Tp = pred.shape
        T = src.shape[1]
        F = min(1 + Tp // HOP, 1 + T // HOP)
        key = (B, Tp, T, length)
        if key not in self._ws:
            f32 = dict(dtype=torch.float32, device=self.device)
            self._ws[key] = dict(frames=torch.empty(B * F, N_FFT, **f32), sp=torch.empty(B * F, P_WIDTH, **f32),
                                 ss=torch.empty(B * F, P_WIDTH, **f32), energy=torch.empty(B, 1025, **f32),
                                 cr=torch.empty(B, dtype=torch.int32, device=self.device),
                                 peak=torch.empty(B, dtype=torch.int32, device=self.device))
        w = self._ws[key]
        L, st = hip.lib(), hip.stream()
        hann = self.c["hann"].data_ptr()
        pred, src = pred.contiguous(), src.contiguous()
        for sig, n, spec in ((pred, Tp, w["sp"]), (src, T, w["ss"])):
            hip.check(L.fh_frame_f32(sig.data_ptr(), hann, w["frames"].data_ptr(), B, n, F, N_FFT, HOP,
                                     N_FFT // 2, 1, st), "fh_frame_f32")
            if _USE_FFT:
                hip.check(L.fh_rfft2048_f32(w["frames"].data_ptr(), self.c["tw"].data_ptr(), spec.data_ptr(), B * F,
                                            0, st), "fh_rfft2048_f32")
            else:
                hip.gemm(w["frames"], self.c["w_fwd"], spec, B * F, P_WIDTH, N_FFT)
        hip.check(L.fh_spec_energy_f32(w["ss"].data_ptr(), w["energy"].data_ptr(), B, F, st), "fh_spec_energy_f32")
        hip.check(L.fh_cutoff_index_f32(w["energy"].data_ptr(), w["cr"].data_ptr(), B, 1025, 0.99, st), "fh_cutoff_index_f32")
        hip.check(L.fh_spec_splice_f32(w["sp"].data_ptr(), w["ss"].data_ptr(), w["cr"].data_ptr(),
                                       w["sp"].data_ptr(), B, F, st), "fh_spec_splice_f32")
        if _USE_FFT:
            hip.check(L.fh_irfft2048_f32(w["sp"].data_ptr(), self.c["tw"].data_ptr(), w["frames"].data_ptr(), B * F, st),
                      "fh_irfft2048_f32")
        else:
            hip.gemm(w["sp"], self.c["w_inv"], w["frames"], B * F, N_FFT, P_WIDTH)
        out = torch.empty(B, length, dtype=torch.float32, device=self.device)
        w["peak"].zero_()
        hip.check(L.fh_istft_ola_f32(w["frames"].data_ptr(), hann, out.data_ptr(), w["peak"].data_ptr(), B, F,
                                     length, N_FFT, HOP, st), "fh_istft_ola_f32")
        hip.check(L.fh_peak_scale_f32(out.data_ptr(), w["peak"].data_ptr(), B, length, 0.99, st), "fh_peak_scale_f32")
        return (out, w["cr"]) if return_cr else out


class Resampler:
    """Device polyphase resampler + peak normalise (the reference does this on the host in numpy)."""

    def __init__(self, device):
        self.device = hip.norm_device(device)
        self._taps = {}

    @hip.on_device
    def __call__(self, x, sr_in, sr_out=48000):
        """x [B, T_in] float32 on device -> [B, T_out], each clip divided by its max |.|."""
        B, n_in = x.shape
        plan = tables.resample_poly_plan(sr_out, sr_in)
        L, st = hip.lib(), hip.stream()
        if plan is None:
            y = x.clone()
        else:
            taps, pre, up, down = plan
            key = (sr_out, sr_in)
            if key not in self._taps:
                self._taps[key] = taps.to(self.device)
            n_out = tables.resample_out_len(n_in, sr_out, sr_in)
            y = torch.empty(B, n_out, dtype=torch.float32, device=self.device)
            x = x.contiguous()
            hip.check(L.fh_resample_poly_f32(x.data_ptr(), self._taps[key].data_ptr(), y.data_ptr(), B, n_in,
                                             n_out, up, down, self._taps[key].numel(), pre, st), "fh_resample_poly_f32")
        peak = torch.zeros(B, dtype=torch.int32, device=self.device)
        hip.check(L.fh_peak_abs_f32(y.data_ptr(), peak.data_ptr(), B, y.shape[1], st), "fh_peak_abs_f32")
        hip.check(L.fh_peak_scale_f32(y.data_ptr(), peak.data_ptr(), B, y.shape[1], 1.0, st), "fh_peak_scale_f32")
        return y
