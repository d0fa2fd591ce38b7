"""Import the REAL reference (/root/reference, Python) on CPU under import shims.

TEST INFRASTRUCTURE ONLY, and only usable in the build container: the GPU box
has no /root/reference.  Used by `oracle/make_golden.py` to produce the vectors
under tests/golden/ and by tests/test_reference_pin.py (skipped when the
reference is absent).

The reference cannot be imported as is: six third-party
modules are missing from this image and CUDA is hard-coded at 11 sites.  The
shims below stand in for *third-party packages* (never for reference code):

  beartype, gateloop_transformer, torchode : decorators / unused classes
  torchdiffeq.odeint   : fixed-grid euler / midpoint (torchdiffeq >= 0.2.3 semantics)
  torchaudio.transforms.Spectrogram / InverseSpectrogram : thin wrappers over
                         torch.stft / torch.istft, which is what torchaudio >= 2.2.1 does
  librosa.filters.mel  : Slaney filter bank (oracle/slaney.py, cross-checked
                         against transformers.audio_utils in tests)
"""
import logging
import os
import sys
import tempfile
import types

import torch

REF_SRC = "/root/reference/src"


def available():
    return os.path.isdir(os.path.join(REF_SRC, "flowhigh"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _odeint(fn, y0, t, *, atol=None, rtol=None, method="midpoint"):
    ys = [y0]
    y = y0
    for i in range(len(t) - 1):
        t0, t1 = t[i], t[i + 1]
        dt = t1 - t0
        if method == "euler":
            y = y + dt * fn(t0, y)
        elif method == "midpoint":
            half = 0.5 * dt
            y = y + dt * fn(t0 + half, y + fn(t0, y) * half)
        else:
            raise ValueError(method)
        ys.append(y)
    return torch.stack(ys)


class _Spectrogram(torch.nn.Module):
    def __init__(self, n_fft, hop_length=None, win_length=None, power=2.0, pad_mode="reflect", **kw):
        super().__init__()
        self.n_fft, self.hop, self.win, self.power, self.pad_mode = n_fft, hop_length, win_length, power, pad_mode
        self.register_buffer("window", torch.hann_window(win_length))

    def forward(self, x):
        s = torch.stft(x, self.n_fft, hop_length=self.hop, win_length=self.win, window=self.window,
                       center=True, pad_mode=self.pad_mode, normalized=False, onesided=True,
                       return_complex=True)
        return s if self.power is None else s.abs().pow(self.power)


class _InverseSpectrogram(torch.nn.Module):
    def __init__(self, n_fft, hop_length=None, win_length=None, pad_mode="reflect", **kw):
        super().__init__()
        self.n_fft, self.hop, self.win = n_fft, hop_length, win_length
        self.register_buffer("window", torch.hann_window(win_length))

    def forward(self, s, length=None):
        return torch.istft(s, self.n_fft, hop_length=self.hop, win_length=self.win,
                           window=self.window, center=True, normalized=False, onesided=True,
                           length=length)


_LOADED = None


def load_reference():
    """-> the `flowhigh` reference package, importable on CPU."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not available():
        raise RuntimeError("reference tree not present")
    import typing
    from .slaney import mel_filter_bank

    ident = lambda f=None, *a, **k: f
    _mod("beartype", beartype=ident)
    sys.modules["beartype.typing"] = typing
    sys.modules["beartype"].typing = typing
    _mod("gateloop_transformer", SimpleGateLoopLayer=type("SimpleGateLoopLayer", (torch.nn.Module,), {}))
    _mod("torchode", Tsit5=object)
    _mod("torchdiffeq", odeint=_odeint)
    ta = _mod("torchaudio")
    ta.transforms = _mod("torchaudio.transforms", Spectrogram=_Spectrogram,
                         InverseSpectrogram=_InverseSpectrogram, MelScale=None, AmplitudeToDB=None)
    ta.functional = _mod("torchaudio.functional", resample=None)
    lr = _mod("librosa")
    lr.filters = _mod("librosa.filters", mel=lambda sr, n_fft, n_mels, fmin, fmax: mel_filter_bank(sr, n_fft, n_mels, fmin, fmax))
    lr.util = _mod("librosa.util", normalize=None)
    lr.resample = None

    # neutralise hard-coded CUDA (flowhighsr.py:122,136; cfm:220-239; init_vocoder.py:14,16; postprocessing.py:7-8)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _orig_load = torch.load

    def _load_cpu(f, *a, **k):
        k.pop("map_location", None)
        k.pop("weights_only", None)
        return _orig_load(f, map_location="cpu", weights_only=False)
    torch.load = _load_cpu

    # the reference writes model_debug.log into CWD at import and formats whole tensors at INFO
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="fh_ref_"))
    try:
        sys.path.insert(0, REF_SRC)
        import flowhigh  # noqa
    finally:
        os.chdir(cwd)
    logging.getLogger().setLevel(logging.WARNING)
    _LOADED = flowhigh
    return flowhigh


def build_reference_model(ckpt_dir, ode_method="euler", cfm_method="basic_cfm", sigma=0.0):
    """FlowHighSR.from_local on CPU, then set solver / probability path like a caller would."""
    fh = load_reference()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = fh.FlowHighSR.from_local(ckpt_dir, "cpu")
    model.odeint_kwargs["method"] = ode_method
    model.set_cfm_method(cfm_method)
    model.sigma = sigma
    return model
