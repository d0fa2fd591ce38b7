"""Real-time factor of the other BASELINE.json configurations on one GPU (bench.py times configs[1] only):
python tools/config_bench.py B secs sr_in method steps [reps]"""
import sys, time
import torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth

B, secs, sr_in, method, steps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 4
dev = torch.device("cuda:0")
cfg = synth.SYNTH_CFG
model = FlowHighSR(FLowHigh(synth.make_state_dict(cfg, 0), cfg, dev), torchdiffeq_ode_method=method,
                   upsampling_method="hip")
n = int(secs * 100)
x = torch.stack([torch.from_numpy(synth.lowres_clip(i, secs, sr_in)) for i in range(B)]).to(dev)
noise = torch.cat([synth.prior_noise(i, n) for i in range(B)], 0).to(dev).contiguous()
for _ in range(2):
    out = model.generate_from_device(x, sr_in, steps, noise=noise)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    out = model.generate_from_device(x, sr_in, steps, noise=noise)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
assert bool(torch.isfinite(out).all())
print(f"B={B} x {secs:g} s, {sr_in}->48000 Hz, {method} x {steps}: {dt * 1e3:.1f} ms per batch = {B * secs / dt:.1f} x real time "
      f"({torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB peak)")
