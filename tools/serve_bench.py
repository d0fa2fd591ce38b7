"""Serving mix: N clips of random lengths (0.5 .. 4 s, 12 kHz) through generate_many -- bucketed by length
(ragged=False: one batch per distinct length) against ONE ragged launch sequence (ragged=True).
python tools/serve_bench.py [n_clips] [profile]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
cfg = synth.SYNTH_CFG
model = FlowHighSR(FLowHigh(synth.make_state_dict(cfg, 0), cfg, dev), torchdiffeq_ode_method="euler",
                   upsampling_method="hip")
rng = np.random.default_rng(0)
lens = [int(rng.integers(5, 41)) * 1200 for _ in range(n)]          # multiples of 0.1 s
clips = [synth.lowres_clip(i, L / 12000, 12000) for i, L in enumerate(lens)]
noise = [synth.prior_noise(i, L * 4 // 480) for i, L in enumerate(lens)]
audio_s = sum(lens) / 12000
ref = [model.generate(c, 12000, noise=z).clone() for c, z in zip(clips, noise)]
for ragged in (False, True, False, True, True):
    model.generate_many(clips, 12000, noise=noise, ragged=ragged)           # plans for every shape
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = model.generate_many(clips, 12000, noise=noise, ragged=ragged)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    same = all(torch.equal(a, b) for a, b in zip(ref, out))
    print(f"{n} clips, {audio_s:.1f} s of audio, {len(set(lens))} lengths, ragged={ragged}: {dt * 1e3:7.1f} ms "
          f"= {audio_s / dt:6.1f} x real time, bit-identical to generate() per clip: {same}")
# a new mix of the same lengths (merged plan rebuilt from cached per-clip plans: the serving case)
perm = rng.permutation(n)
c2, z2 = [clips[i] for i in perm], [noise[i] for i in perm]
torch.cuda.synchronize()
t = time.perf_counter()
model.generate_many(c2, 12000, noise=z2, ragged=True)
torch.cuda.synchronize()
print(f"new order of the same clips (merge rebuilt): {(time.perf_counter() - t) * 1e3:7.1f} ms")
if len(sys.argv) > 2:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    model.generate_many(clips, 12000, noise=noise, ragged=True)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
