"""Randomised end-to-end check: generate_batch on random clip lengths / rates / batch sizes / solvers (full-width
SYNTH-CFG vocoder: every plan-time choice - tile shapes, split-K, fused / unfused closing convs, phase-major
layouts - depends on the shape) against the CPU oracle.  python tests/tools/e2e_fuzz.py [n_cases] [seed]"""
import random
import sys
import torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth
from oracle import ref_cpu

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(min(16, torch.get_num_threads()))
cfg = synth.SYNTH_CFG
sd = synth.make_state_dict(cfg, 0)
fh = FLowHigh(sd, cfg, "cuda")
worst = 0.0
for case in range(n_cases):
    sr = rng.choice([8000, 12000, 16000, 24000])
    n_in = rng.choice([rng.randint(sr // 100 + 1, sr // 5), rng.randint(sr // 5, sr), rng.randint(sr, 3 * sr)])
    B = rng.choice([1, 1, 2, 3])
    method, steps = rng.choice([("euler", 1), ("midpoint", 1), ("euler", 2)])
    m = FlowHighSR(fh, torchdiffeq_ode_method=method, upsampling_method=rng.choice(["scipy", "hip"]))
    clips = [synth.lowres_clip(500 + 10 * case + i, n_in / sr, sr)[:n_in] for i in range(B)]
    t48 = -(-len(clips[0]) * 48000 // sr)
    n = t48 // 480
    if n < 1 or t48 <= 784:           # (reflect padding of 784 samples: shorter clips fail in the reference's torch.stft as well)
        continue
    noise = torch.cat([synth.prior_noise(500 + 10 * case + i, n) for i in range(B)], 0)
    out = m.generate_batch(clips, sr, 48000, steps, noise=noise).cpu()
    err = 0.0
    for i in range(B if B == 1 else 2):          # the oracle is the slow part: first two clips of a batch
        ref = ref_cpu.generate(sd, cfg, clips[i], sr, noise[i:i + 1], steps, method)
        err = max(err, (out[i:i + 1] - ref).abs().max().item())
    worst = max(worst, err)
    ok = err <= 1e-4 and bool(torch.isfinite(out).all())
    print(f"{'ok  ' if ok else 'FAIL'} case {case}: sr={sr} n_in={len(clips[0])} ({len(clips[0]) / sr:.3f} s) B={B} {method} x{steps} "
          f"upsampling={m.upsampling_method} err={err:.2e}", flush=True)
print(f"{n_cases} cases, worst error {worst:.2e}")
