"""CPU: repository contract checks -- the C-ABI library builds/loads and exports every declared
symbol, the product never imports the oracle, constant tables agree with the oracle's."""
import ctypes
import os
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
    # F(5,4): rows of 2^24 - 4096 samples or more are refused before anything is launched (the kernel finds a sample's phase
    # in fp32: include/flowhigh_hip.h); a fake non-null descriptor pointer is enough for the argument checks
    assert lib.fh_conv_wino54_f32(16, 1, 1, 96, (1 << 24) - 4096, 1, 0, 1, 0) == -1
    assert b"fh_conv_wino54_f32" in lib.fh_last_error() and b"rows of" in lib.fh_last_error()
    assert lib.fh_conv_wino54_f32(16, 1, 1, 100, 1000, 1, 0, 1, 0) == -1          # cout_pad not a multiple of the 96-row tile
    # the two bf16 x 6 entries of ABI 5: channel count / dilation / flags of the narrow-stage conv, K and alignment of the GEMM
    assert lib.fh_narrow_tile_len() == 256
    assert lib.fh_narrow_conv_bf16x6_f32(16, 1, 16, 1, 50, 1, 1, 0) == -1 and b"channels" in lib.fh_last_error()
    assert lib.fh_narrow_conv_bf16x6_f32(16, 1, 16, 1, 24, 7, 1, 0) == -1 and b"dilation" in lib.fh_last_error()
    assert lib.fh_narrow_conv_bf16x6_f32(16, 1, 16, 1, 24, 1, 2, 0) == -1 and b"flags" in lib.fh_last_error()
    assert lib.fh_gemm_bf16x6_f32(16, 32, 16, 0, 0, 0, 16, 64, 8, 64, 40, 1.0, 0, 0) == -1 and b"multiple of 32" in lib.fh_last_error()
    assert lib.fh_gemm_bf16x6_f32(16, 30, 16, 0, 0, 0, 16, 64, 8, 64, 64, 1.0, 0, 0) == -1 and b"aligned" in lib.fh_last_error()


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: only tests/ (incl. tests/tools/), __graft_entry__.smoke() and the
    # cpu_baseline leg of bench.py may touch it
    for f in list((ROOT / "flowhigh_amd").rglob("*.py")) + list((ROOT / "tools").glob("*.py")):
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


# ------------------------------------------------------------------------------------------
# Winograd host side: weight transform / packing and the phase-major layout (no GPU needed)
# ------------------------------------------------------------------------------------------
_BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                    [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
_AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
                   dtype=torch.float64)


@pytest.mark.parametrize("c,k", [(16, 3), (32, 7), (48, 11), (16, 5)])
def test_pack_wino_weight_is_the_f43_transform_of_the_conv(c, k):
    """The packed layout [ci/16, G, 6, cout_pad, 16] with B^T / A^T as the kernel applies them (conv_wino.hip)
    reproduces F.conv1d: y[4t + i] = sum_xi A^T[i, xi] sum_{g, ci} U[g, xi, co, ci] (B^T x[4t + 3g - center ..])[xi]."""
    import torch.nn.functional as F
    from flowhigh_amd import vocoder as V
    g = torch.Generator().manual_seed(k)
    w = torch.randn(c, c, k, generator=g, dtype=torch.float64)
    x = torch.randn(1, c, 64, generator=g, dtype=torch.float64)
    cpad = 64
    u = V.pack_wino_weight(w, cpad).double()                      # [c/16, G, 6, cpad, 16]
    ng, center = -(-k // 3), (k - 1) // 2
    assert tuple(u.shape) == (c // 16, ng, 6, cpad, 16)
    assert float(u[:, :, :, c:].abs().max()) == 0.0               # padded output rows
    xp = torch.nn.functional.pad(x, (center, 3 * ng + 8))
    tiles = 64 // 4
    m = torch.zeros(6, cpad, tiles, dtype=torch.float64)
    for gi in range(ng):
        d = xp[0, :, 3 * gi:].unfold(-1, 6, 4)[:, :tiles]         # [c, tiles, 6]
        v = torch.einsum("xj,ctj->xct", _BT, d)
        ug = u[:, gi].permute(1, 2, 0, 3).reshape(6, cpad, c)     # [6, cpad, ci]
        m += torch.einsum("xoc,xct->xot", ug, v)
    y = torch.einsum("ix,xot->oti", _AT, m).reshape(cpad, 64)[:c]
    ref = F.conv1d(x, w, padding=center)[0]
    # U is stored in float32: the error is the float32 rounding of the transformed weights
    assert (y - ref).abs().max().item() <= 1e-5 * float(ref.abs().max())


# F(5,4): rows of B^T as conv_wino54.hip applies them (kB8Off / kB8Coef: v = x[unit] + sum_j coef_j x[off_j]) and A^T (the sums /
# differences of its epilogue); points 0, 1, -1, 2, -2, 1/2, -1/2, inf
_B8_OFF = [[2, 4, 0, 0, 0, 6]] + [[1, 2, 3, 4, 5, 6]] * 6 + [[1, 3, 5, 1, 1, 7]]
_B8_COEF = [[5.25, -5.25, -1, 0, 0], [1, 1, -4.25, -4.25, 1], [-1, 1, 4.25, -4.25, -1], [.5, .25, -2.5, -1.25, 2],
            [-.5, .25, 2.5, -1.25, -2], [2, 4, -2.5, -5, .5], [-2, 4, 2.5, -5, -.5], [-1, 5.25, -5.25, 0, 0]]
_AT8 = torch.tensor([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, .5, -.5, 0], [0, 1, 1, 4, 4, .25, .25, 0],
                     [0, 1, -1, 8, -8, .125, -.125, 0], [0, 1, 1, 16, 16, .0625, .0625, 1]], dtype=torch.float64)


@pytest.mark.parametrize("c,k", [(16, 3), (32, 7), (48, 11), (16, 5), (16, 12)])
def test_pack_wino54_weight_is_the_f54_transform_of_the_conv(c, k):
    """pack_wino54_weight ([ci/16, G, 8, cout_pad, 16], taps in groups of 4) with the B^T rows and A^T of conv_wino54.hip
    reproduces F.conv1d: y[5t + i] = sum_xi A^T[i, xi] sum_{g, ci} U[g, xi, co, ci] (B^T x[5t + 4g - center ..])[xi]; and the
    constants are the Toom-Cook matrices of tests/tools/winograd_numerics.py for the same points."""
    import importlib.util
    import torch.nn.functional as F
    from flowhigh_amd import vocoder as V
    bt = torch.zeros(8, 8, dtype=torch.float64)
    for xi in range(8):
        bt[xi, _B8_OFF[xi][5]] += 1.0
        for j in range(5):
            bt[xi, _B8_OFF[xi][j]] += _B8_COEF[xi][j]
    spec = importlib.util.spec_from_file_location("wn", ROOT / "tests" / "tools" / "winograd_numerics.py")
    wn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wn)
    from fractions import Fraction as Fr
    at_ref, g_ref, bt_ref = wn.toom_cook(5, 4, [0, 1, -1, 2, -2, Fr(1, 2), -Fr(1, 2)])
    assert torch.equal(bt, bt_ref) and torch.equal(_AT8, at_ref)
    assert (torch.tensor(V._WINO54_G, dtype=torch.float64) - g_ref).abs().max().item() < 1e-15
    g = torch.Generator().manual_seed(k)
    w = torch.randn(c, c, k, generator=g, dtype=torch.float64)
    x = torch.randn(1, c, 60, generator=g, dtype=torch.float64)
    cpad = 64
    u = V.pack_wino54_weight(w, cpad).double()                    # [c/16, G, 8, cpad, 16]
    ng, center = -(-k // 4), (k - 1) // 2
    assert tuple(u.shape) == (c // 16, ng, 8, cpad, 16) and float(u[:, :, :, c:].abs().max()) == 0.0
    xp = torch.nn.functional.pad(x, (center, 4 * ng + 8))
    tiles = 60 // 5
    m = torch.zeros(8, cpad, tiles, dtype=torch.float64)
    for gi in range(ng):
        d = xp[0, :, 4 * gi:].unfold(-1, 8, 5)[:, :tiles]         # [c, tiles, 8]
        v = torch.einsum("xj,ctj->xct", bt, d)
        ug = u[:, gi].permute(1, 2, 0, 3).reshape(8, cpad, c)     # [8, cpad, ci]
        m += torch.einsum("xoc,xct->xot", ug, v)
    y = torch.einsum("ix,xot->oti", _AT8, m).reshape(cpad, 60)[:c]
    xr = F.pad(x, (center, k - 1 - center))                       # (even k: one more sample on the right)
    ref = F.conv1d(xr, w)[0]
    assert (y - ref).abs().max().item() <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("c,co,ci,k", [(24, 24, 24, 11), (48, 48, 48, 7), (32, 32, 32, 3), (40, 33, 40, 5), (8, 8, 5, 1), (16, 16, 16, 11)])
def test_pack_narrow_bf_weight_is_the_conv_in_k_blocks(c, co, ci, k):
    """pack_narrow_bf_weight as narrow_bf.hip reads it: per slab (narrow_slabs) and k-block, lane l of [N tile][piece] holds the
    8 input channels of pair q = 4 kb + (l >> 4) = tap og + octet for output channel 16 n + (l & 15); the three bf16 pieces sum to
    the fp32 weight EXACTLY, padding pairs / rows / channels are zero, and the size is what the kernel's cursor advances by."""
    from flowhigh_amd import packing as P
    g = torch.Generator().manual_seed(k + c)
    w = torch.randn(co, ci, k, generator=g)
    u = P.pack_narrow_bf_weight(w, c)
    ma = -(-c // 16)
    slabs = P.narrow_slabs(c)
    assert sum(og for _, og in slabs) == c // 8 and all(og <= 3 for _, og in slabs)
    assert u.numel() * 4 == sum(-(-(k * og) // 4) for _, og in slabs) * ma * 3 * 1024
    pieces = u.view(torch.int16).view(torch.bfloat16).float()
    back, pos = torch.zeros(16 * ma, c, k), 0
    for ob, og in slabs:
        nb = -(-(k * og) // 4)
        blk = pieces[pos:pos + nb * ma * 3 * 512].view(nb, ma, 3, 64, 8)
        pos += nb * ma * 3 * 512
        full = blk[:, :, 0] + blk[:, :, 1] + blk[:, :, 2]             # exact: h + m + l is the fp32 value
        for kb in range(nb):
            for lg in range(4):
                q = 4 * kb + lg
                vals = full[kb, :, 16 * lg:16 * lg + 16]              # [na, n, 8]
                if q >= k * og:
                    assert float(vals.abs().max()) == 0.0
                    continue
                tap, o = divmod(q, og)
                back[:, 8 * (ob + o):8 * (ob + o) + 8, tap] = vals.reshape(16 * ma, 8)
    assert torch.equal(back[:co, :ci], w) and float(back[co:].abs().max() if co < 16 * ma else 0.0) == 0.0
    assert float(back[:, ci:].abs().max() if ci < c else 0.0) == 0.0


@pytest.mark.parametrize("L,d,pm", [(5000, 5, 1), (5000, 3, 1), (20000, 5, 1), (250, 5, 1), (13, 3, 1), (1, 1, 1), (5000, 5, 0), (777, 1, 0)])
def test_wino54_tile_counts_host_and_library_agree(L, d, pm):
    """vocoder.wino_n_tiles (run maps, launch model) = fh_wino54_n_tiles (the launcher): phase-major rows are tiled as one
    sequence of 5-output tile slots, every phase its ceil(n / 5) tiles + >= 3 empty ones, rounded up to a multiple of 4."""
    from flowhigh_amd import hip, vocoder as V
    got = V.wino_n_tiles(V.WINO_F54, L, d, pm)
    assert got == hip.lib().fh_wino54_n_tiles(L, d, pm)
    n = -(-L // d)
    if pm:
        slots = (-(-n // 5) + 3 + 3) // 4 * 4
        assert got == -(-d * slots // 64) and slots >= -(-n // 5) + 3
    else:
        assert got == -(-n // 320) * d


@pytest.mark.parametrize("L,d", [(23, 3), (1000, 5), (17, 1), (4, 5)])
def test_phase_major_round_trip(L, d):
    from flowhigh_amd import hip, vocoder as V
    x = torch.randn(2, 3, L)
    pm = V.to_phase_major(x, d)
    lp = V.phase_len(L, d)
    assert lp == hip.lib().fh_phase_len(L, d) and lp % 4 == 0 and d * lp >= L
    assert tuple(pm.shape) == (2, 3, d * lp)
    for t in (0, L // 2, L - 1):
        assert torch.equal(pm[..., (t % d) * lp + t // d], x[..., t])
    assert torch.equal(V.from_phase_major(pm, d, L), x)


def test_wino_rule_and_tiles():
    from flowhigh_amd import vocoder as V
    assert V.pick_wino_tile(768)[0] & 1 == 0 and V.pick_wino_tile(768)[1] == 768
    assert V.pick_wino_tile(96) == (1, 96)
    assert V.pick_wino_tile(48)[1] == 64
    assert V.use_wino(768, 5) and V.use_wino(192, 3) and not V.use_wino(24, 1)


def test_wino_tile_choice_follows_the_launch_model():
    """Plan-time tile choice (vocoder.choose_wino_cfg): only shapes the packed cout_pad allows, small launches
    go to small tiles, a one-group launch with too few 64 x 512 blocks is cut finer, and at large batch the
    default shape stays (hysteresis)."""
    from flowhigh_amd import vocoder as V
    ks = lambda c: [c // 16 * g for g in (4, 3, 1)]                    # k = 11 / 7 / 3 residual stacks
    assert V.choose_wino_cfg(ks(768), 1, 768, 500, 1, 0)[0] == 5       # 0.5 s clip: 36 big blocks -> 32 x 256
    assert V.choose_wino_cfg(ks(96), 1, 96, 120000, 1, 1)[0] == 1      # cout_pad 96: only 96- and 32-row tiles
    assert V.choose_wino_cfg(ks(64), 1, 64, 240000, 1, 0)[0] in (0, 4, 5)
    assert V.choose_wino_cfg([sum(ks(192))], 1, 192, 60000, 1, 0)[0] != 0   # 354 blocks = 1.4 rounds of 256 CUs
    assert V.choose_wino_cfg(ks(192), 32, 192, 60000, 1, 0)[0] == 0       # large batch: within 3 %, default stays
    assert V.choose_wino_cfg(ks(384), 32, 384, 20000, 1, 0)[0] in (0, 6)  # (128-row tile: ~8 % fewer issue slots)
    for cfg in (0, 1, 4, 5, 6):                                        # cost grows with the work
        assert V.wino_launch_cost(ks(384), 2, 384, 20000, 1, cfg) > V.wino_launch_cost(ks(384), 1, 384, 20000, 1, cfg)
    # one block on one CU = its own model time
    a, b = V._WINO_COST[0]
    assert abs(V.wino_launch_cost([10], 1, 64, 512, 1, 0) - (a * 10 + b) * (1 + 0.12 / 200)) < 1e-6


def test_split_k_is_for_short_clips_only(monkeypatch):
    """vocoder.wino_split_k: input-channel slices where ONE block's K loop bounds the launch (clips under ~2 s at
    C = 768), none for the 10 s headline shape; off with FH_WINO_SPLITK=0; never more slices than divide C / 16."""
    from flowhigh_amd import vocoder as V
    monkeypatch.delenv("FH_WINO_SPLITK", raising=False)
    ks = [11, 7, 3]
    assert V.wino_split_k(ks, 768, 768, 250, 1, 0) == 3          # 0.5 s clip, first stage
    assert V.wino_split_k(ks, 768, 768, 500, 1, 0) >= 2          # 1 s
    for c, up in ((768, 5), (384, 20), (192, 60), (96, 120)):
        for d in (1, 3, 5):
            assert V.wino_split_k(ks, c, V.pick_wino_tile(c)[1], 1000 * up, d, V.pick_wino_tile(c)[0]) == 1
    assert V.wino_split_k(ks, 80, 128, 100, 1, 0) == 1            # 80 / 16 = 5 chunks: neither 2 nor 3 slices
    monkeypatch.setenv("FH_WINO_SPLITK", "0")
    assert V.wino_split_k(ks, 768, 768, 250, 1, 0) == 1


def test_shape_cache_is_bounded_by_bytes_and_count():
    """hip.ShapeCache: LRU over launch plans / workspaces, bounded by the device bytes they hold."""
    from flowhigh_amd.hip import ShapeCache
    c = ShapeCache(max_entries=3, max_bytes=4000)
    mk = lambda n: dict(a=torch.empty(n, dtype=torch.uint8), keep=[torch.empty(n, dtype=torch.uint8)])
    c["x"] = mk(500)
    c["y"] = mk(500)
    _ = c["x"]                                  # x is now the most recent
    c["z"] = mk(500)
    assert list(c) == ["y", "x", "z"]
    c["w"] = mk(500)                            # count bound: the least recent goes
    assert list(c) == ["x", "z", "w"]
    c["big"] = mk(1600)                         # 3200 bytes: only 800 more fit
    assert list(c) == ["big"] or list(c) == ["w", "big"]
    assert sum(c._bytes[k] for k in c) <= 4000
    c["huge"] = mk(5000)                        # larger than the budget: kept alone
    assert list(c) == ["huge"]


def test_committed_bench_line_follows_the_contract():
    """profiles/r02+_bench_line_B1.json is a bench.py output line: the driver's keys, the two roofline objects and
    cpu_baseline, and self-consistent numbers (a roofline fraction is a fraction: <= 1)."""
    import json
    files = [f for f in sorted((ROOT / "profiles").glob("r*_bench_line_B1.json")) if f.name >= "r02"]
    if not files:
        pytest.skip("no bench line of the current format committed yet")
    line = json.loads(files[-1].read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_hbm", "cpu_baseline"):
        assert k in line, k
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    form = line["config"].get("conv_form", "winograd")              # (lines before round 6 have no conv_form: the fp32-MFMA form)
    assert line["data"] == "synthetic" and "workload" in line["config"]
    # dtype = the arithmetic the path computes in: plain "f32", or -- the bf16 x 6 form -- f32 in / out / accumulate spelled out
    assert line["dtype"] == "f32" if form != "bf16x6" else (line["dtype"].startswith("f32 in / out / accumulate") and "bf16" in line["dtype"])
    assert "model" not in line["config"]
    assert line["steps"] >= 100
    rl = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert k in rl, k
    # the dominant kernel family on ITS matrix instructions: fp32 MFMA peak, or the dense bf16 peak for the bf16 x 6 form
    assert rl["bound"] == "mfma" and rl["unit"] == "TFLOP/s" and rl["peak"] == (2500.0 if form == "bf16x6" else 157.3)
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    assert 0.0 < rl["frac"] <= 1.0                                # issued to the matrix cores / that pipe's peak
    if "by_family" in rl:                                         # round 6 on
        fam = rl["by_family"]
        assert next(iter(fam)).startswith("wino54") and all(0.0 < f["frac"] <= 1.0 for f in fam.values())
        assert rl["all_conv"]["executed_fp32_equiv_tflops"] <= rl["all_conv"]["algorithmic_equiv"]
    else:
        assert rl["achieved"] <= rl["algorithmic_equiv"]          # Winograd executes fewer FLOPs than the direct form
    assert rl["conv_ms_per_step"] <= line["ms_per_step"]
    rh = line["roofline_hbm"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rh, k
    assert rh["bound"] == "hbm" and rh["unit"] == "GB/s" and rh["peak"] == 8000.0
    assert abs(rh["frac"] - rh["achieved"] / rh["peak"]) < 1e-3 and 0.0 < rh["frac"] <= 1.0
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["unit"] == line["unit"] and "10 s clip" in cb["sample"]
    secs_per_step = line["ms_per_step"] / 1e3
    assert abs(line["value"] - line["n_gpus"] * 10.0 / secs_per_step) / line["value"] < 0.02      # B = 1, 10 s clips


def test_bench_self_launches_multi_gpu_runs_before_touching_a_gpu():
    """`python bench.py --gpus 2` without a launcher must start torch.distributed.run as a child (never exec, never
    after a GPU call).  No GPU here: the parent must stop at the device count with a clear message, rc != 0."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr
    src = (ROOT / "bench.py").read_text()
    assert "os.exec" not in src and "execv" not in src
    assert src.index("self_launch(args)") < src.index("torch.cuda.set_device")


def test_integration_stub_descriptor_matches_the_library():
    """The ctypes stub INTEGRATION.md shows a reference maintainer must describe fh_act_group as the library lays it
    out: its _fields_ are parsed from the document and sized against fh_sizeof_act_group()."""
    from flowhigh_amd import hip
    text = (ROOT / "INTEGRATION.md").read_text()
    m = re.search(r"class _ActGroup\(ctypes\.Structure\):.*?\n\s*_fields_ = (\[.*?\])\s*(#[^\n]*)?\n", text, re.S)
    assert m, "INTEGRATION.md no longer shows the _ActGroup stub"
    fields = eval(re.sub(r"#[^\n]*", "", m.group(1)), {"ctypes": ctypes})        # noqa: S307 (our own document)
    stub = type("_ActGroup", (ctypes.Structure,), {"_fields_": fields})
    assert ctypes.sizeof(stub) == hip.lib().fh_sizeof_act_group() == ctypes.sizeof(hip.ActGroup)
    assert [f[0] for f in fields] == [f[0] for f in hip.ActGroup._fields_]
