import sys, time, torch
sys.path.insert(0, '.')
from flowhigh_amd import synth
from oracle import ref_cpu
cfg = synth.SYNTH_CFG
sd = synth.make_state_dict(cfg, 0)
audio = synth.lowres_clip(0, 4.0, 12000); noise = synth.prior_noise(0, 400)
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    ref_cpu.generate(sd, cfg, synth.lowres_clip(1, 0.5, 12000), 12000, synth.prior_noise(1, 50), 1, "euler")
    t = time.perf_counter(); ref_cpu.generate(sd, cfg, audio, 12000, noise, 1, "euler"); dt = time.perf_counter() - t
    print(f"threads {th}: 4 s clip in {dt:.2f} s = {4.0 / dt:.2f} x real time", flush=True)
