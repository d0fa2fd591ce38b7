"""Per-launch table of the vocoder's MERGED (ragged) plan for the serving mix of tools/serve_bench.py:
kind, groups, params, us.  Run on the GPU box: python tools/ragged_layers.py [n_clips]"""
import sys, torch
import numpy as np
sys.path.insert(0, '.')
from flowhigh_amd import hip, synth
from flowhigh_amd.vocoder import Vocoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(0)
frames = [int(rng.integers(5, 41)) * 10 for _ in range(n)]
cfg = synth.SYNTH_CFG
voc = Vocoder(cfg, synth.make_state_dict(cfg, 0), 'cuda:0')
rp = voc.plan_ragged(frames)
for _ in range(3):
    voc.run_ragged(rp)
torch.cuda.synchronize()
steps = rp["steps"]
acc = [0.0] * len(steps)
R = 5
L, base = hip.lib(), rp["desc"].data_ptr()
for _ in range(R):
    evs = []
    for s in steps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        voc.run_ragged(dict(rp, steps=[s]))
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(evs):
        acc[i] += a.elapsed_time(b) * 1e3 / R
tot = {}
for s, t in zip(steps, acc):
    tot[s[0]] = tot.get(s[0], 0.0) + t
    print(f"{s[0]:6s} {str(s[2:]):60s} {t:9.1f} us")
print({k: round(v / 1e3, 3) for k, v in tot.items()}, "ms; total", round(sum(acc) / 1e3, 3), "ms for", sum(frames) / 100, "s of audio")
