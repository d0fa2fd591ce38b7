"""Bitwise comparison of the experimental Winograd block shapes (tile 8 = tools/exp/conv_wino2.hip, tile 9 =
tools/exp/conv_wino3.hip) with the product's 64 x 256 tile.  Needs the experiment build:
    tools/exp/build_wino_variants.sh
    FH_LIB_PATH=flowhigh_amd/lib/abl/winox.so python -m pytest tools/exp/test_wino_variants.py -q"""
import sys
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from flowhigh_amd import hip, vocoder as V      # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


@pytest.mark.parametrize("k,d,pm,nres,L", [(11, 1, False, 1, 2000), (7, 5, True, 0, 1501), (3, 1, False, 3, 512),
                                           (11, 3, True, 0, 96), (1, 1, False, 0, 260)])
def test_conv_wino2_equals_the_twelve_wave_kernel_bitwise(k, d, pm, nres, L):
    """conv_wino2.hip (tile 8) against conv_wino.hip's 64 x 256 tile (tile 4) on the same descriptors: bias, up to
    three residuals, scale, phase-major rows, batch 3, a channel count that is not a multiple of the tile (cout 80 in
    cout_pad 128): same bits."""
    c, cpad, B = 80, 128, 3
    x, w, b = rnd(B, c, L, seed=300), rnd(c, c, k, seed=301, scale=1.0 / (c * k) ** 0.5), rnd(c, seed=302)
    res = [rnd(B, c, L, seed=303 + i) for i in range(nres)]
    conv = lambda t: (V.to_phase_major(t, d) if pm else t).to(DEV)
    xd, rd = conv(x), [conv(r) for r in res]
    ud, bd = V.pack_wino_weight(w, cpad).to(DEV), b.to(DEV)
    outs = []
    for cfg in (4, 8, 9):
        out = torch.full_like(xd, float("nan"))
        g = V.make_wino_group([V.make_wino_seg(xd, ud, c, k)], bd, rd, out, c, cpad, L, scale=0.5)
        keep = V.conv_wino([g], B, cpad, L, d, DEV, cfg, phase_major=pm)
        torch.cuda.synchronize()
        outs.append(V.from_phase_major(out.cpu(), d, L) if pm else out.cpu())
        del keep
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = ((F.conv1d(x.double(), w.double(), b.double(), dilation=d, padding=(k - 1) // 2 * d)
            + sum(r.double() for r in res)) * 0.5).float()
    assert maxdiff(outs[1], ref) <= 3e-5


def test_conv_wino2_three_segments_and_transposed_phases():
    """Tile 8 on the other two launch forms: the fused stage-closing conv (three K segments of 4 / 3 / 1 tap groups,
    three residuals, / 3) and the output phases of a ConvTranspose1d (strided stores)."""
    c, L, B = 64, 1204, 2
    ks = [11, 7, 3]
    xs = [rnd(B, c, L, seed=320 + i) for i in range(3)]
    ws = [rnd(c, c, k, seed=330 + i, scale=0.05) for i, k in enumerate(ks)]
    rs = [rnd(B, c, L, seed=350 + i) for i in range(3)]
    bsum = rnd(c, seed=340)
    ref = (sum(F.conv1d(x, w, None, padding=(k - 1) // 2) + r for x, w, r, k in zip(xs, ws, rs, ks)) + bsum.view(1, -1, 1)) / 3
    xd, rd, bd = [x.to(DEV) for x in xs], [r.to(DEV) for r in rs], bsum.to(DEV)
    ud = [V.pack_wino_weight(w, c).to(DEV) for w in ws]
    outs = []
    for cfg in (4, 8, 9):
        out = torch.full((B, c, L), float("nan"), device=DEV)
        g = V.make_wino_group([V.make_wino_seg(xd[i], ud[i], c, k) for i, k in enumerate(ks)], bd, rd, out, c, c, L,
                              scale=1.0 / 3)
        keep = V.conv_wino([g], B, c, L, 1, DEV, cfg)
        torch.cuda.synchronize()
        outs.append(out.cpu())
        del keep
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and maxdiff(outs[1], ref) <= 2e-5
    u, k, cin, cout, L = 4, 8, 64, 64, 1000
    x, wt, b = rnd(B, cin, L, seed=360), rnd(cin, cout, k, seed=361, scale=0.2), rnd(cout, seed=362)
    ref = F.conv_transpose1d(x.double(), wt.double(), b.double(), stride=u, padding=(k - u) // 2).float()
    xd, bd = x.to(DEV), b.to(DEV)
    out = torch.full((B, cout, u * L), float("nan"), device=DEV)
    groups, keep = [], []
    for r, taps in enumerate(V.transposed_conv_phases(k, u)):
        w, center = V.wino_phase_weight(wt, taps)
        ud1 = V.pack_wino_weight(w, cout).to(DEV)
        keep.append(ud1)
        groups.append(V.make_wino_group([V.make_wino_seg(xd, ud1, cin, w.shape[-1], center)], bd, [], out, cout, cout, L,
                                        stride=u, phase=r))
    for cfg in (8, 9, 9):           # (tile 9 twice: the second launch finds the cursors the first one left zeroed)
        out.fill_(float("nan"))
        keep.append(V.conv_wino(groups, B, cout, L, 1, DEV, cfg))
        torch.cuda.synchronize()
        assert maxdiff(out, ref) <= 2e-5
