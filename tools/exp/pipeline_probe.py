"""Would pipelining consecutive generate() calls on two streams pay?  Front end + transformer of call i + 1 on a side stream
while the vocoder + post-processing of call i run on the main stream (the small kernels fill the tails of the conv launches).
Timing experiment only: python tools/exp/pipeline_probe.py [steps]"""
import sys, time, torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
cfg = synth.SYNTH_CFG
m = FlowHighSR(FLowHigh(synth.make_state_dict(cfg, 0), cfg, dev), torchdiffeq_ode_method="euler", upsampling_method="hip")
x = torch.from_numpy(synth.lowres_clip(0, 10.0, 12000))[None].to(dev)
z = synth.prior_noise(0, 1000).to(dev).contiguous()
voc = m.flowhigh.vocoder


def serial():
    return m.generate_from_device(x, 12000, 1, noise=z)


main = torch.cuda.current_stream()
side = torch.cuda.Stream()
consumed = torch.cuda.Event()
consumed.record(main)


def piped():
    global consumed
    with torch.cuda.stream(side):
        side.wait_event(consumed)                      # the previous call's vocoder has copied its mel
        cond = m.resampler(x, 12000, 48000)
        mel = m._sample(cond=cond, time_steps=1, cfm_method=m.cfm_method, noise=z, decode_to_audio=False)
        ready = torch.cuda.Event(); ready.record(side)
    cond.record_stream(main)
    main.wait_event(ready)
    p = voc.plan(1, mel.shape[1])
    p["mel_in"][:, :voc.true_mels].copy_(mel.transpose(1, 2))
    consumed = torch.cuda.Event(); consumed.record(main)
    voc.run(p)
    return m.postproc(p["wav"], cond, cond.size(-1))


ref = serial().clone()
for name, fn in (("serial", serial), ("two streams", piped), ("serial", serial), ("two streams", piped)):
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    print(f"{name:12s}: {dt * 1e3:.3f} ms per step = {10 / dt:.1f} x real time; same bits as serial: {torch.equal(out, ref)}")
