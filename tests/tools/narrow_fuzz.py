"""Randomised check of the narrow-stage conv kernels against float64 F.conv1d through the C ABI: channels 8 .. 48, odd kernels
1 .. 11, dilations 1 .. 6, batches, ragged group lengths (aligned or not), 1-3 K segments with their own inputs, 0-3 residuals,
bias or none, scale.   python tests/tools/narrow_fuzz.py [n_cases] [seed] [direct | winograd]
direct: fh_narrow_conv_bf16x6_f32 (narrow_bf.hip), tolerance 5e-6 of the output scale; winograd: fh_amp_actconv_f32, 3e-5."""
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from flowhigh_amd import vocoder as V        # noqa: E402

DEV = "cuda:0"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
direct = not (len(sys.argv) > 3 and sys.argv[3] == "winograd")
pack = V.pack_narrow_bf_weight if direct else V.pack_amp_weight
tol = 5e-6 if direct else 3e-5
worst, fails = 0.0, 0
for case in range(n_cases):
    C = rng.choice([8, 16, 24, 24, 32, 40, 48, 48])
    d = rng.choice([1, 1, 2, 3, 4, 5, 6])
    B = rng.choice([1, 1, 2, 3])
    ngroups = rng.choice([1, 2, 3])
    nseg = rng.choice([1, 1, 1, 2, 3])
    ks = [rng.choice([1, 3, 5, 7, 9, 11]) for _ in range(nseg)]
    if not direct:                                      # the Winograd kernel places all segments in one slab: kernel sizes not too far apart
        cmax = max((k - 1) // 2 for k in ks)
        if any(cmax - (k - 1) // 2 + 4 * -(-k // 4) + 3 > 16 for k in ks):
            ks = [max(ks)] * nseg
    g = torch.Generator().manual_seed(1000 + case)
    groups, keep, refs, outs = [], [], [], []
    for gi in range(ngroups):
        L = rng.choice([rng.randint(1, 40), rng.randint(41, 700), rng.randint(701, 5000), 4 * rng.randint(1, 900)])
        nres = rng.choice([0, 1, 1, 2, 3])
        scale = rng.choice([1.0, 1.0, 0.5, 1.0 / 3.0])
        has_bias = rng.random() < 0.8
        bias = torch.randn(C, generator=g) if has_bias else None
        total = torch.zeros(B, C, L, dtype=torch.float64)
        segs = []
        for k in ks:
            x = torch.randn(B, C, L, generator=g) * rng.choice([0.01, 1.0, 1.0, 30.0])
            w = torch.randn(C, C, k, generator=g) / (C * k) ** 0.5
            total += F.conv1d(x.double(), w.double(), None, dilation=d, padding=(k - 1) // 2 * d)
            xd, ud = x.to(DEV), pack(w, C).to(DEV)
            keep += [xd, ud]
            segs.append(V.make_amp_seg(xd, ud, k, direct=direct))
        res = [torch.randn(B, C, L, generator=g) for _ in range(nres)]
        ref = total + (bias.double()[None, :, None] if has_bias else 0.0) + sum(r.double() for r in res)
        refs.append(ref * scale)
        rd = [r.to(DEV) for r in res]
        bd = bias.to(DEV) if has_bias else None
        out = torch.full((B, C, L), float("nan"), device=DEV)
        keep += rd + [bd]
        outs.append(out)
        groups.append(V.make_amp_group(segs, bd, rd, out, L, scale=scale, direct=direct))
    keep.append(V.amp_actconv(groups, B, C, d, DEV, direct=direct))
    torch.cuda.synchronize()
    for gi, (o, r) in enumerate(zip(outs, refs)):
        err = float((o.cpu().double() - r).abs().max()) / max(1.0, float(r.abs().max()))
        if not err <= tol:
            fails += 1
            print(f"FAIL case {case} group {gi}: C={C} ks={ks} d={d} B={B} L={o.shape[-1]} err={err}")
        worst = max(worst, err if err == err else float("inf"))
print(f"{n_cases} cases ({'direct bf16 x 6' if direct else 'Winograd fp32'} narrow-stage kernel), {fails} failures, worst error {worst:.2e} of the output scale (tolerance {tol:g})")
sys.exit(1 if fails else 0)
