"""Slaney-scale mel filter bank (oracle side; TEST INFRASTRUCTURE ONLY).

Restates the published algorithm of `librosa.filters.mel` (librosa >= 0.9.2,
unpinned in /root/reference/pyproject.toml:8; librosa itself is NOT in the
reference tree nor in this image).  Call site in the reference:
/root/reference/src/flowhigh/models/melvoco.py:64-70
    librosa_mel_fn(sr=48000, n_fft=2048, n_mels=256, fmin=20, fmax=24000)
i.e. htk=False (Slaney scale), norm='slaney', dtype float32.

Pinned by tests/test_oracle_cpu.py against
`transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney')`,
an independent implementation that ships in this image.
"""
import numpy as np

_F_SP = 200.0 / 3.0
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    mel = f / _F_SP
    log_t = f >= _MIN_LOG_HZ
    safe = np.where(log_t, f, _MIN_LOG_HZ)
    return np.where(log_t, _MIN_LOG_MEL + np.log(safe / _MIN_LOG_HZ) / _LOGSTEP, mel)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f = m * _F_SP
    log_t = m >= _MIN_LOG_MEL
    return np.where(log_t, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), f)


def mel_filter_bank(sr=48000, n_fft=2048, n_mels=256, fmin=20.0, fmax=24000.0):
    """-> float32 [n_mels, 1 + n_fft//2], triangular filters, area ('slaney') normalised."""
    n_bins = 1 + n_fft // 2
    fft_f = np.linspace(0.0, sr / 2.0, n_bins)
    mel_pts = np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2)
    hz_pts = mel_to_hz(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)
