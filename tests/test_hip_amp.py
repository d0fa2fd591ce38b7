"""GPU: the narrow-stage conv kernels (the AMP-block convs of the stages with <= 48 channels: amp_fused.hip = fp32-MFMA Winograd
F(5,4), narrow_bf.hip = direct bf16 x 6, parameter `direct`) against the float64 definition of the conv, through the C ABI.  (The form with the Activation1d inside the launch, and its tests against the oracle's
activation, left with ABI 4: `git show 60fcf48:tests/test_hip_amp.py`.)"""
import sys
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowhigh_amd import hip                 # noqa: E402,F401
from flowhigh_amd import vocoder as V        # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def pack(w, C, direct):
    return (V.pack_narrow_bf_weight if direct else V.pack_amp_weight)(w, C)


FORMS = pytest.mark.parametrize("direct", [False, True], ids=["winograd_f32", "direct_bf16x6"])


def conv_ref(a, w, d):
    k = w.shape[-1]
    return F.conv1d(a.double(), w.double(), None, dilation=d, padding=(k - 1) // 2 * d)


@pytest.mark.parametrize("C,k,d,L,B", [(24, 11, 1, 1000, 1), (24, 7, 3, 1201, 2), (48, 11, 5, 2000, 1), (48, 3, 1, 644, 2),
                                       (8, 7, 1, 333, 1), (16, 11, 3, 900, 1), (32, 5, 2, 1280, 1), (40, 9, 4, 777, 1),
                                       (48, 7, 6, 1500, 1), (24, 3, 5, 7, 1), (24, 11, 5, 24, 2), (48, 11, 1, 320, 1)])
@FORMS
def test_amp_conv_only(C, k, d, L, B, direct):
    """The conv alone against float64, bias + residual + scale."""
    x, w, b = rnd(B, C, L, seed=1), rnd(C, C, k, seed=2, scale=1.0 / (C * k) ** 0.5), rnd(C, seed=3)
    r1 = rnd(B, C, L, seed=4)
    ref = ((conv_ref(x, w, d) + b.double()[None, :, None] + r1.double()) * 0.5).float()
    xd, rd, bd = x.to(DEV), r1.to(DEV), b.to(DEV)
    out = torch.full_like(xd, float("nan"))
    ud = pack(w, C, direct).to(DEV)
    g = V.make_amp_group([V.make_amp_seg(xd, ud, k, direct=direct)], bd, [rd], out, L, scale=0.5, direct=direct)
    keep = V.amp_actconv([g], B, C, d, DEV, direct=direct)
    torch.cuda.synchronize()
    assert maxdiff(out.cpu(), ref) <= (4e-6 if direct else 2e-5)
    del keep


@pytest.mark.parametrize("C,k,d,L,B", [(24, 11, 1, 1000, 1), (24, 7, 3, 1201, 2), (48, 11, 5, 2000, 1), (8, 3, 1, 5, 1),
                                       (32, 7, 2, 1283, 1), (24, 3, 1, 1, 1)])
@FORMS
def test_amp_three_groups_in_one_launch(C, k, d, L, B, direct):
    """Three groups (the AMP blocks of a stage) with their own weights, inputs and residuals in one launch."""
    groups, keep, refs, outs = [], [], [], []
    for j in range(3):
        x, w, b = rnd(B, C, L, seed=20 + j, scale=1.5), rnd(C, C, k, seed=30 + j, scale=1.0 / (C * k) ** 0.5), rnd(C, seed=40 + j)
        r1 = rnd(B, C, L, seed=50 + j)
        refs.append((conv_ref(x, w, d) + b.double()[None, :, None] + r1.double()).float())
        xd, rd, bd, ud = x.to(DEV), r1.to(DEV), b.to(DEV), pack(w, C, direct).to(DEV)
        out = torch.full_like(xd, float("nan"))
        keep += [xd, rd, bd, ud]
        outs.append(out)
        groups.append(V.make_amp_group([V.make_amp_seg(xd, ud, k, direct=direct)], bd, [rd], out, L, direct=direct))
    keep.append(V.amp_actconv(groups, B, C, d, DEV, direct=direct))
    torch.cuda.synchronize()
    for j in range(3):
        assert maxdiff(outs[j].cpu(), refs[j]) <= 3e-5


@FORMS
def test_amp_three_segments_fused_average(direct):
    """The stage-closing position: one group, three K segments (k = 11 / 7 / 3 on three inputs), three residuals,
    scale 1 / 3 (models.py:181-187)."""
    C, L, B = 24, 1100, 2
    segs, keep, total = [], [], 0.0
    res = [rnd(B, C, L, seed=70 + j) for j in range(3)]
    bias = rnd(C, seed=60)
    for j, k in enumerate((3, 11, 7)):
        x, w = rnd(B, C, L, seed=80 + j, scale=1.5), rnd(C, C, k, seed=90 + j, scale=1.0 / (C * k) ** 0.5)
        total = total + conv_ref(x, w, 1)
        xd, ud = x.to(DEV), pack(w, C, direct).to(DEV)
        keep += [xd, ud]
        segs.append(V.make_amp_seg(xd, ud, k, direct=direct))
    ref = ((total + bias.double()[None, :, None] + sum(r.double() for r in res)) / 3.0).float()
    rd = [r.to(DEV) for r in res]
    out = torch.full((B, C, L), float("nan"), device=DEV)
    bd = bias.to(DEV)            # (descriptors hold raw pointers: every tensor they name must outlive the launch)
    g = V.make_amp_group(segs, bd, rd, out, L, scale=1.0 / 3.0, direct=direct)
    keep.append(V.amp_actconv([g], B, C, 1, DEV, direct=direct))
    torch.cuda.synchronize()
    assert maxdiff(out.cpu(), ref) <= 3e-5


def test_amp_refuses_the_form_that_left_with_abi_4():
    """flags bit 1 clear asked for the Activation1d inside the launch: FH_E_ARG with a message, not a silent conv."""
    C, k, L = 24, 3, 320
    xd, ud = rnd(1, C, L, seed=1).to(DEV), V.pack_amp_weight(rnd(C, C, k, seed=2), C).to(DEV)
    out = torch.empty_like(xd)
    g = hip.to_device_struct_array([V.make_amp_group([V.make_amp_seg(xd, ud, k)], None, [], out, L)], DEV)
    tiles = V.amp_tile_list([L], 1, 1).to(DEV)
    rc = hip.lib().fh_amp_actconv_f32(g.data_ptr(), 1, tiles.data_ptr(), tiles.shape[0], C, 1, 1, 1, hip.stream())
    assert rc != 0 and "left the library" in hip.lib().fh_last_error().decode()


@FORMS
@pytest.mark.parametrize("d", [1, 3, 5])
def test_amp_ragged_groups_and_alignment_give_the_same_bits(d, direct):
    """Groups of different lengths in one launch (ragged batches), a clip inside a batch, and rows that are not 16-byte
    aligned (4-byte accesses): every clip gets the bits of its own single-group, aligned launch."""
    C, k = 24, 7
    w, b = rnd(C, C, k, seed=6, scale=0.1), rnd(C, seed=7)
    ud, bd = pack(w, C, direct).to(DEV), b.to(DEV)
    lens = [1203, 320, 2000, 17]
    xs = [rnd(1, C, L, seed=200 + i, scale=1.5).to(DEV) for i, L in enumerate(lens)]

    def run(items, batch=1):
        outs = [torch.full_like(x, float("nan")) for x in items]
        gs = [V.make_amp_group([V.make_amp_seg(x, ud, k, direct=direct)], bd, [x], o, x.shape[-1], direct=direct) for x, o in zip(items, outs)]
        keep = V.amp_actconv(gs, batch, C, d, DEV, direct=direct)
        torch.cuda.synchronize()
        del keep
        return outs

    alone = [run([x])[0] for x in xs]
    together = run(xs)
    for a, t in zip(alone, together):
        assert torch.equal(a, t)
    # the same clip as item 1 of a batch of 2
    xb = torch.cat([xs[0].flip(-1), xs[0]], dim=0).contiguous()
    assert torch.equal(run([xb], batch=2)[0][1:], alone[0])
    # a chunk of the clip that starts on a block boundary of every dilation: its inner samples keep their bits
    # (the direct form's bits do not depend on where a chunk starts at all: any start)
    s = 1203 if direct else 1200
    tail = run([xs[2][..., s:].contiguous()])[0]
    halo = ((k - 1) // 2 + 5) * d             # taps + the other inputs of an F(5,4) tile (rounding)
    assert torch.equal(tail[..., halo:], alone[2][..., s + halo:])


@pytest.mark.parametrize("form", ["direct", "winograd"])
def test_narrow_kernels_randomised_configurations(form):
    """tests/tools/narrow_fuzz.py as a test: 150 random (channels, taps, dilation, batch, ragged lengths, K segments, residuals,
    bias, scale) launches of each narrow-stage kernel against float64 F.conv1d (5e-6 direct bf16 x 6, 3e-5 Winograd fp32)."""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "tests" / "tools" / "narrow_fuzz.py"), "150", "11", form], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "FAIL" not in r.stdout and "150 cases" in r.stdout and " 0 failures" in r.stdout, r.stdout[-2000:]
