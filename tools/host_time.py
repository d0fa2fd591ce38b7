import sys, time, torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth
dev = torch.device("cuda:0")
cfg = synth.SYNTH_CFG
m = FlowHighSR(FLowHigh(synth.make_state_dict(cfg, 0), cfg, dev), torchdiffeq_ode_method="euler", upsampling_method="hip")
x = torch.from_numpy(synth.lowres_clip(0, 10.0, 12000))[None].to(dev)
z = synth.prior_noise(0, 1000).to(dev).contiguous()
for _ in range(5): m.generate_from_device(x, 12000, 1, noise=z)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50): m.generate_from_device(x, 12000, 1, noise=z)
host = (time.perf_counter() - t) / 50
torch.cuda.synchronize()
tot = (time.perf_counter() - t) / 50
print(f"host enqueue time per step {host*1e3:.2f} ms, wall per step {tot*1e3:.2f} ms")
