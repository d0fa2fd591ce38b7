"""CPU: repository contract checks -- the C-ABI library builds/loads and exports every declared
symbol, the product never imports the oracle, constant tables agree with the oracle's."""
import ctypes
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    from flowhigh_amd import build, hip
    build.build(verbose=False)
    lib = hip.lib()
    header = (ROOT / "include" / "flowhigh_hip.h").read_text()
    declared = set(re.findall(r"\b(fh_[a-z0-9_]+)\s*\(", header))
    declared -= {"fh_conv_seg", "fh_conv_group", "fh_act_group"}
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/flowhigh_hip.h but not exported"
    assert set(hip.EXPORTS) == declared
    assert lib.fh_sizeof_conv_group() == ctypes.sizeof(hip.ConvGroup)
    assert lib.fh_sizeof_act_group() == ctypes.sizeof(hip.ActGroup)


def test_argument_errors_are_reported_not_crashed():
    from flowhigh_amd import hip
    lib = hip.lib()
    assert lib.fh_gemm_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 0, 0) == -1
    assert b"fh_gemm_f32" in lib.fh_last_error()
    assert lib.fh_conv_grouped_f32(0, 1, 1, 128, 10, 9, 8, 0) == -1


def test_product_never_imports_oracle():
    for f in list((ROOT / "flowhigh_amd").rglob("*.py")):
        txt = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f"{f} imports the oracle"
    code = "import sys; import flowhigh_amd, flowhigh_amd.synth; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_product_has_no_cpu_fallback():
    from flowhigh_amd import FLowHigh, hip, synth
    with pytest.raises(hip.HipError):
        FLowHigh({}, synth.TINY_CFG, "cpu")


def test_tables_agree_with_oracle_and_torch():
    from flowhigh_amd import tables
    from oracle import slaney
    assert np.abs(tables.slaney_mel_basis() - slaney.mel_filter_bank()).max() < 1e-7
    # DFT-by-GEMM weights reproduce torch.stft / torch.istft on CPU (fp64 check of the tables)
    g = torch.Generator().manual_seed(0)
    frames = torch.randn(5, 2048, generator=g, dtype=torch.float64)
    wf = tables.dft_forward_weight().double()
    spec = (frames @ wf.T)[:, :2112]                                 # P-layout
    ref = torch.fft.rfft(frames)
    f = np.arange(1025)
    re_i, im_i = (f // 32) * 64 + f % 32, (f // 32) * 64 + f % 32 + 32
    assert (spec[:, re_i] - ref.real).abs().max() < 1e-4 and (spec[:, im_i] - ref.imag).abs().max() < 1e-4
    back = spec @ tables.dft_inverse_weight()[:2048].double().T
    assert (back - frames).abs().max() < 1e-4
    plan = tables.resample_poly_plan(48000, 12000)
    assert plan[2:] == (4, 1) and plan[1] == 41


def test_weight_packing_roundtrip():
    from flowhigh_amd import vocoder as V
    w = torch.arange(24 * 16 * 3, dtype=torch.float32).view(24, 16, 3)
    p = V.pack_conv_weight(w, 32, 8)
    assert p.shape == (2, 3, 32, 8)
    assert p[1, 2, 5, 3] == w[5, 11, 2] and p[:, :, 24:].abs().sum() == 0
    for u, k in [(5, 11), (4, 8), (3, 7), (2, 4), (8, 16)]:
        taps = V.transposed_conv_phases(k, u)
        assert sorted(j for ph in taps for j, _ in ph) == list(range(k))
    sd = {"a.weight_g": torch.tensor([[[2.0]], [[3.0]]]), "a.weight_v": torch.ones(2, 4, 1), "a.bias": torch.zeros(2)}
    f = V.fold_weight_norm(sd)
    assert torch.allclose(f["a.weight"], torch.tensor([1.0, 1.5]).view(2, 1, 1).expand(2, 4, 1))
