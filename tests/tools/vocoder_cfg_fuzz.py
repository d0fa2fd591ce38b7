"""Randomised vocoder CONFIGURATIONS (round 5: which kernel a stage runs on depends on its channel count, kernel sizes and
dilations: narrow-stage kernel / F(5,4) / F(4,3) / direct, fused or per-phase upsamplers): random upsample rates, initial
channels, kernel sizes, dilations, block type and activation, each checked three ways: Vocoder.forward against the oracle
(oracle/ref_cpu.py: bigvgan_forward), the time-chunked run and the ragged run against the plain run bit for bit.
    python tests/tools/vocoder_cfg_fuzz.py [n_cases] [seed]          (GPU box)"""
import random
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from flowhigh_amd import synth               # noqa: E402
from flowhigh_amd import vocoder as V        # noqa: E402
from oracle import ref_cpu                   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)
RATES = [[8, 6, 5, 2], [5, 4, 3, 2, 2, 2], [6, 5, 4, 4], [10, 8, 6], [5, 4, 4, 3, 2], [4, 4, 3, 5, 2], [8, 5, 4, 3]]
worst, fails = 0.0, 0
for case in range(n_cases):
    rates = rng.choice(RATES)
    kernels = [u * rng.choice([1, 2, 2]) + rng.choice([0, 0, 0, 1]) * (u > 2) for u in rates]       # k = u, 2u, sometimes odd k - u
    c0 = rng.choice([64, 96, 128, 192, 256, 320, 384, 768]) if len(rates) <= 4 else rng.choice([256, 384, 512, 768, 1536])
    nk = rng.choice([1, 2, 3, 3, 3, 4])
    ks = sorted(rng.sample([3, 5, 7, 9, 11], min(nk, 5)))
    nm = rng.choice([1, 2, 3])
    shared = rng.random() < 0.7                      # the blocks share their dilation lists (else mixed-dilation positions)
    dl = [rng.choice([1, 2, 3, 4, 5, 6, 7, 9]) for _ in range(nm)]
    dils = [list(dl) if shared else [rng.choice([1, 2, 3, 5]) for _ in range(nm)] for _ in ks]
    cfg = dict(synth.SYNTH_CFG, upsample_rates=rates, upsample_kernel_sizes=kernels, upsample_initial_channel=c0,
               resblock=rng.choice(["1", "1", "2"]), resblock_kernel_sizes=ks, resblock_dilation_sizes=dils,
               activation=rng.choice(["snakebeta", "snake"]), snake_logscale=rng.random() < 0.7)
    if c0 // 2 ** len(rates) < 1:
        continue
    sd = synth.make_vocoder_state_dict(cfg, seed=case)
    try:
        voc = V.Vocoder(cfg, sd, "cuda:0")
    except NotImplementedError as e:
        print(f"skip case {case}: {e}")
        continue
    halo, align = voc.chunk_geometry()
    B, N = rng.choice([1, 2]), rng.choice([rng.randint(3, 40), rng.randint(40, 160)])
    g = torch.Generator().manual_seed(case)
    mel = (torch.randn(B, N, 256, generator=g) * 2.0 - 3.0)
    ref = ref_cpu.bigvgan_forward(sd, cfg, mel.transpose(1, 2)).squeeze(1)
    wav = voc.forward(mel.cuda()).clone()
    err = float((wav.cpu() - ref).abs().max())
    kinds = {s[0] for s in voc.plan(B, N)["steps"]}
    notes = []
    if N > align + halo:                             # chunked == whole (several chunks)
        if not torch.equal(voc.forward_chunked(mel.cuda(), align), wav):
            notes.append("CHUNKED DIFFERS")
    lens = [N, max(1, N // 3), max(2, N - 1)]
    outs = voc.forward_ragged([mel[0, :n].cuda().contiguous() for n in lens])
    outs = [o.clone() for o in outs]
    for n, o in zip(lens, outs):
        if not torch.equal(o, voc.forward(mel[:1, :n].cuda())):
            notes.append(f"RAGGED DIFFERS n={n}")
    ok = err <= 2e-5 * max(1.0, float(ref.abs().max())) and not notes and bool(torch.isfinite(wav).all())
    worst, fails = max(worst, err), fails + (not ok)
    chans = [st["c"] for st in voc.stages]
    print(f"{'ok  ' if ok else 'FAIL'} case {case}: rates={rates} k={kernels} c0={c0} chans={chans} ks={ks} dil={dils} rb={cfg['resblock']} "
          f"{cfg['activation']}{'/log' if cfg['snake_logscale'] else ''} B={B} N={N} align={align} kinds={sorted(kinds)} err={err:.2e} {' '.join(notes)}", flush=True)
print(f"{n_cases} cases, {fails} failed, worst error {worst:.2e}")
