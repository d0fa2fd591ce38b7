"""Randomised check of fh_act1d_grouped_pm_f32 (plain / phase-major in and out, ragged lengths, several groups)
against the oracle's Activation1d.  python tests/tools/act_fuzz.py [n_cases] [seed] [wide]
wide: phase-major dilations up to 16 on either side and rows of up to 20 000 samples (several tiles per row at every dilation)."""
import sys, random, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, synth, vocoder as V
from oracle import ref_cpu
DEV = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
filt = synth.kaiser_sinc_filter()
worst = 0.0
for case in range(n_cases):
    B, C, G = rng.choice([1, 2, 3]), rng.choice([1, 3, 8]), rng.choice([1, 2, 3])
    L = rng.choice([rng.randint(1, 30), rng.randint(31, 1100), rng.randint(1101, 5000)])
    din, dout = rng.choice([(1, 1), (1, 3), (3, 1), (1, 5), (5, 1), (2, 1), (1, 2), (3, 5)])
    if WIDE:
        d = rng.choice([4, 6, 7, 9, 11, 13, 16])
        din, dout = rng.choice([(1, d), (d, 1), (d, rng.choice([3, 8, 16]))])
        L = rng.choice([rng.randint(1, 200), rng.randint(900, 2200), rng.randint(2201, 20000)])
    kind = rng.choice(["snakebeta_log", "snake_lin"])
    g = torch.Generator().manual_seed(1000 + case)
    keep, refs, groups, outs = [], [], [], []
    for gi in range(G):
        x = torch.randn(B, C, L, generator=g) * 1.5
        al, be = torch.randn(C, generator=g) * 0.4, torch.randn(C, generator=g) * 0.4
        if kind == "snakebeta_log":
            h = {"activation": "snakebeta", "snake_logscale": True}
            sd = {"a.act.alpha": al, "a.act.beta": be}
            alpha, beta = torch.exp(al), torch.exp(be)
        else:
            h = {"activation": "snake", "snake_logscale": False}
            al = al.abs() + 0.5
            sd = {"a.act.alpha": al}
            alpha, beta = al, al
        sd["a.upsample.filter"] = filt
        sd["a.downsample.lowpass.filter"] = filt
        refs.append(ref_cpu.activation1d(sd, "a.", x, h))
        p = dict(alpha=alpha.to(DEV), inv_beta=(1.0 / (beta + 1e-9)).to(DEV), up=filt.flatten().tolist(), down=filt.flatten().tolist())
        xd = (V.to_phase_major(x, din) if din > 1 else x).to(DEV)
        yd = torch.full((B, C, dout * V.phase_len(L, dout) if dout > 1 else L), float("nan"), device=DEV)
        keep += [p, xd]
        outs.append(yd)
        groups.append(V.make_act_group(xd, yd, p))
    keep.append(V.act1d_grouped(groups, B, C, L, DEV, din, dout))
    torch.cuda.synchronize()
    for gi in range(G):
        got = V.from_phase_major(outs[gi].cpu(), dout, L) if dout > 1 else outs[gi].cpu()
        err = (got - refs[gi]).abs().max().item()
        worst = max(worst, err)
        if not (err <= 4e-6 and bool(torch.isfinite(got).all())):
            print(f"FAIL case {case}: B={B} C={C} G={G} L={L} din={din} dout={dout} {kind} group {gi} err={err}")
print(f"{n_cases} cases, worst error {worst:.2e}")
