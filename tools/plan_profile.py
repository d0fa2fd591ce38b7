"""Where the time of the FIRST call for a new clip length goes (launch-plan build).  python tools/plan_profile.py"""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth

dev = torch.device("cuda:0")
cfg = synth.SYNTH_CFG
model = FlowHighSR(FLowHigh(synth.make_state_dict(cfg, 0), cfg, dev), torchdiffeq_ode_method="euler",
                   upsampling_method="hip")
def run(secs):
    x = torch.from_numpy(synth.lowres_clip(0, secs, 12000))[None].to(dev)
    noise = synth.prior_noise(0, int(secs * 100)).to(dev).contiguous()
    torch.cuda.synchronize(); t = time.perf_counter()
    model.generate_from_device(x, 12000, 1, noise=noise)
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
run(1.0)
for secs in (1.3, 2.1, 3.7):
    print(f"{secs} s clip: first call {run(secs):7.1f} ms, second call {run(secs):6.1f} ms")
pr = cProfile.Profile(); pr.enable(); run(2.9); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
