"""One-off: BASELINE configs[1] (10 s clip, 12 -> 48 kHz, euler x 1, SYNTH-CFG) HIP vs the CPU oracle."""
import sys, time, torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth
from oracle import ref_cpu
torch.set_num_threads(16)
cfg = synth.SYNTH_CFG
sd = synth.make_state_dict(cfg, 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
for method, sr in (("euler", 12000), ("midpoint", 16000)):
    m = FlowHighSR(FLowHigh(sd, cfg, "cuda"), torchdiffeq_ode_method=method)
    audio = synth.lowres_clip(0, secs, sr)
    noise = synth.prior_noise(0, int(secs * 100))
    out, st = m.generate_batch([audio], sr, 48000, 1, noise=noise, return_stages=True)
    t = time.time()
    ref, rs = ref_cpu.generate(sd, cfg, audio, sr, noise, 1, method, return_stages=True)
    print(f"{method} {sr}->48k {secs}s: oracle {time.time()-t:.1f}s  cr {int(st['cr'][0])}=={rs['cr']}  "
          f"wav max-abs {(st['wav'].cpu()-rs['wav']).abs().max():.3e}  out max-abs {(out.cpu()-ref).abs().max():.3e}", flush=True)
