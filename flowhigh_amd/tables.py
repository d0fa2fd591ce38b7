"""Constant tables of the path, built once on the host (float64 -> float32) and kept resident in HBM.

  mel basis       librosa.filters.mel(sr=48000, n_fft=2048, n_mels=256, fmin=20, fmax=24000), Slaney
                  scale + area norm (call site /root/reference/src/flowhigh/models/melvoco.py:64-70)
  DFT bases       2048-point real DFT / inverse DFT as GEMM operands in the packed "P-layout" that
                  the GEMM pair-epilogue and the spectrum kernels share (include/flowhigh_hip.h)
  hann            torch.hann_window(2048) (periodic), exactly as the reference builds it
  rotary table    cos/sin of fl32(pos) * inv_freq, built with torch CPU fp32 ops like
                  models/pos_emb.py:44-51 (an fp64-derived table breaks parity)
  resample taps   scipy.signal.resample_poly's kaiser FIR (flowhighsr.py:68)
"""
import math
from functools import lru_cache

import numpy as np
import torch

N_FFT = 2048
HOP = 480
N_BINS = 1025
N_MELS = 256
P_BLOCKS = 33
P_WIDTH = P_BLOCKS * 64       # 2112
MAG_WIDTH = P_BLOCKS * 32     # 1056


def _pad_rows(w, mult=128):
    n = w.shape[0]
    n_pad = (n + mult - 1) // mult * mult
    if n_pad == n:
        return w.contiguous()
    out = torch.zeros((n_pad,) + tuple(w.shape[1:]), dtype=w.dtype)
    out[:n] = w
    return out


def slaney_mel_basis(sr=48000, n_fft=N_FFT, n_mels=N_MELS, fmin=20.0, fmax=24000.0):
    """[n_mels, 1 + n_fft // 2] float32.  Vectorised; independent of oracle/slaney.py."""
    def hz2mel(f):
        f = np.asarray(f, np.float64)
        lin = f * 3.0 / 200.0
        log = 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4))
        return np.where(f >= 1000.0, log, lin)

    def mel2hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * (np.log(6.4) / 27.0)), m * 200.0 / 3.0)

    edges = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    freqs = np.arange(1 + n_fft // 2, dtype=np.float64) * (sr / n_fft)
    lo, ce, hi = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    up = (freqs[None, :] - lo) / (ce - lo)
    down = (hi - freqs[None, :]) / (hi - ce)
    tri = np.clip(np.minimum(up, down), 0.0, None)
    tri *= (2.0 / (hi - lo))
    return tri.astype(np.float32)


@lru_cache(maxsize=None)
def mel_gemm_weight():
    """W [256, 1056]: mel basis against the MAG epilogue output (bins >= 1025 are zero columns)."""
    w = torch.zeros(N_MELS, MAG_WIDTH)
    w[:, :N_BINS] = torch.from_numpy(slaney_mel_basis())
    return _pad_rows(w)


@lru_cache(maxsize=None)
def hann_window():
    return torch.hann_window(N_FFT)


def _p_index():
    """bin f -> (packed column of Re, packed column of Im)."""
    f = np.arange(N_BINS)
    re = (f // 32) * 64 + (f % 32)
    return re, re + 32


@lru_cache(maxsize=None)
def dft_forward_weight():
    """W [2112 -> padded 2176, 2048]: row p of the P-layout holds cos / -sin of its bin, so that
    frames[., 2048] @ W^T = packed (Re, Im) spectrum == torch.stft(onesided)."""
    k = np.arange(N_FFT, dtype=np.float64)
    f = np.arange(N_BINS, dtype=np.float64)
    ang = 2.0 * np.pi * ((f[:, None] * k[None, :]) % N_FFT) / N_FFT
    w = np.zeros((P_WIDTH, N_FFT), dtype=np.float32)
    re, im = _p_index()
    w[re] = np.cos(ang)
    w[im] = -np.sin(ang)
    return _pad_rows(torch.from_numpy(w))


@lru_cache(maxsize=None)
def dft_inverse_weight():
    """W [2048, 2112]: packed spectrum @ W^T = irfft (C2R: Im of DC / Nyquist ignored, 1/N scale)."""
    k = np.arange(N_FFT, dtype=np.float64)
    f = np.arange(N_BINS, dtype=np.float64)
    ang = 2.0 * np.pi * ((k[:, None] * f[None, :]) % N_FFT) / N_FFT
    c = np.full(N_BINS, 2.0)
    c[0] = 1.0
    c[-1] = 1.0
    w = np.zeros((N_FFT, P_WIDTH), dtype=np.float32)
    re, im = _p_index()
    w[:, re] = (np.cos(ang) * c[None, :] / N_FFT)
    sn = -np.sin(ang) * c[None, :] / N_FFT
    sn[:, 0] = 0.0
    sn[:, -1] = 0.0
    w[:, im] = sn
    return _pad_rows(torch.from_numpy(w))


@lru_cache(maxsize=None)
def fft_twiddles():
    """[1024, 2] float64: (cos, -sin)(2 pi k / 2048) (fh_rfft2048_f32 / fh_irfft2048_f32)."""
    k = np.arange(N_FFT // 2, dtype=np.float64)
    ang = 2.0 * np.pi * k / N_FFT
    return torch.from_numpy(np.stack([np.cos(ang), -np.sin(ang)], axis=1)).contiguous()


def rotary_tables(inv_freq, n):
    """cos, sin [n, 32] float32 from the checkpoint's fp32 inv_freq (pos_emb.py:44-51)."""
    inv_freq = inv_freq.detach().float().cpu()
    t = torch.arange(n).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return freqs.cos().contiguous(), freqs.sin().contiguous()


@lru_cache(maxsize=None)
def resample_poly_plan(up, down):
    """Restates the filter design and alignment of scipy.signal.resample_poly (scipy >= 1.10,
    window=('kaiser', 5.0), padtype='constant') -> (taps float32 incl. pre-padding, n_pre_remove,
    up, down) with up/down reduced by their gcd."""
    from scipy.signal import firwin
    g = math.gcd(up, down)
    up, down = up // g, down // g
    if up == 1 and down == 1:
        return None
    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(np.float32)
    h = h * np.float32(up)
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    taps = np.concatenate([np.zeros(n_pre_pad, np.float32), h])
    return torch.from_numpy(taps), n_pre_remove, up, down


def resample_out_len(n_in, up, down):
    g = math.gcd(up, down)
    up, down = up // g, down // g
    n = n_in * up
    return n // down + bool(n % down)
