"""End-to-end time of a 10 s clip with the k = 2 u upsamplers on the odd rates of hop 480 (synth.ODD_CFG at 1536 channels: stage
lengths 5 N + 1, 20 N + 4, 80 N + 16, 240 N + 49, 480 N + 98 -- rows that are not 16-byte aligned at three of five stages) against
SYNTH-CFG, with the per-kernel split of one step.  python tools/odd_cfg_bench.py"""
import sys, time, torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth
dev = torch.device("cuda:0")
for name, cfg in (("SYNTH_CFG", synth.SYNTH_CFG), ("ODD_CFG x 1536", dict(synth.ODD_CFG, upsample_initial_channel=1536))):
    m = FlowHighSR(FLowHigh(synth.make_state_dict(cfg, 0), cfg, dev), torchdiffeq_ode_method="euler", upsampling_method="hip")
    x = torch.from_numpy(synth.lowres_clip(0, 10.0, 12000))[None].to(dev)
    z = synth.prior_noise(0, 1000).to(dev).contiguous()
    voc = m.flowhigh.vocoder
    for _ in range(3):
        m.generate_from_device(x, 12000, 1, noise=z)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        m.generate_from_device(x, 12000, 1, noise=z)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    voc.conv_timing, voc.act_timing = [], []
    m.generate_from_device(x, 12000, 1, noise=z); torch.cuda.synchronize()
    conv = sum(a.elapsed_time(b) for a, b in voc.conv_timing); act = sum(a.elapsed_time(b) for a, b in voc.act_timing)
    nconv, nact = len(voc.conv_timing), len(voc.act_timing)
    voc.conv_timing = voc.act_timing = None
    flops = voc.conv_flops_per_frame() * 1000
    print(f"{name}: {dt * 1e3:.2f} ms per 10 s clip = {10 / dt:.0f} x real time; conv {nconv} launches {conv:.2f} ms "
          f"({flops / conv / 1e9:.0f} TFLOP/s direct-form), activations {nact} launches {act:.2f} ms")
