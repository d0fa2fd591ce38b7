"""Eager vs HIP-graph latency of generate_from_device for short clips.  python tools/graph_latency.py"""
import sys, time, torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth
cfg = synth.SYNTH_CFG
m = FlowHighSR(FLowHigh(synth.make_state_dict(cfg, 0), cfg, "cuda"), torchdiffeq_ode_method="euler", upsampling_method="hip")
for secs in (0.5, 1.0, 2.0, 5.0, 10.0):
    n_in = int(secs * 12000)
    g = m.capture(1, n_in, 12000, 1)
    x = torch.from_numpy(synth.lowres_clip(0, secs, 12000))[None].cuda()
    noise = synth.prior_noise(0, int(secs * 100)).cuda().reshape(int(secs * 100), -1).contiguous()
    g.x.copy_(x); g.noise.copy_(noise)
    def timeit(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
    te = timeit(lambda: m.generate_from_device(x, 12000, 1, noise=noise))
    tg = timeit(g.replay)
    same = torch.equal(g.replay(), m.generate_from_device(x, 12000, 1, noise=noise))
    print(f"{secs:5.1f} s clip: eager {te:7.3f} ms  graph {tg:7.3f} ms  ({secs * 1e3 / tg:6.0f} x real time)  identical {same}")
