"""Measured maxima of the device log-mel (frontend.LogMel: fh_frame_f32 -> fh_rfft2048_f32 -> fh_gemm_f32[LOGCLAMP]) against the
oracle's fp32 torch.stft log-mel (oracle/ref_cpu.py: logmel) and against its float64 run, split as SURVEY.md 8a asks: mel > -8
(signal) and below (near the log(1e-5) clamp, where the REFERENCE'S OWN fp32 FFT noise is amplified by the log).
    python tests/tools/logmel_maxima.py            (GPU box)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from flowhigh_amd.frontend import LogMel    # noqa: E402
from oracle import ref_cpu                  # noqa: E402

lm = LogMel("cuda:0")
print(f"{'case':28s} {'hip-ref32 loud':>15s} {'hip-ref32 all':>14s} {'hip-ref64 loud':>15s} {'ref32-ref64 loud':>17s} {'ref32-ref64 all':>16s}")
for name, n, scale, seed in (("noise 0.1, 0.2 s", 9600, 0.1, 5), ("noise 0.1, 2 s", 96000, 0.1, 6), ("noise 1.0, 1 s", 48000, 1.0, 7),
                             ("noise 0.01, 1 s", 48000, 0.01, 8), ("12 kHz band-limited, 2 s", 96000, 0.1, 9)):
    g = torch.Generator().manual_seed(seed)
    audio = torch.randn(2, n, generator=g) * scale
    if "band" in name:      # a low-rate clip upsampled 4x: the high band is near-silent, as in the workload
        import scipy.signal
        low = torch.randn(2, n // 4, generator=g).numpy() * scale
        audio = torch.from_numpy(scipy.signal.resample_poly(low, 4, 1, axis=-1).astype("float32"))
    r32 = ref_cpu.logmel(audio)
    r64 = ref_cpu.logmel(audio.double()).float()
    mel = lm(audio.cuda()).view(r32.shape).cpu()
    loud = r64 > -8.0
    d32, d64, dr = (mel - r32).abs(), (mel - r64).abs(), (r32 - r64).abs()
    f = lambda t, m: float(t[m].max()) if m.any() else 0.0
    print(f"{name:28s} {f(d32, loud):15.2e} {float(d32.max()):14.2e} {f(d64, loud):15.2e} {f(dr, loud):17.2e} {float(dr.max()):16.2e}")
