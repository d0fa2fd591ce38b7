import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import FLowHigh, FlowHighSR, synth
cfgname, B, method, sr, secs = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), float(sys.argv[5])
cfg = getattr(synth, cfgname)
sd = synth.make_state_dict(cfg, 0)
m = FlowHighSR(FLowHigh(sd, cfg, 'cuda'), torchdiffeq_ode_method=method, upsampling_method='hip')
clips = [synth.lowres_clip(i, secs, sr) for i in range(B)]
n = int(secs * 100)
noise = torch.cat([synth.prior_noise(i, n) for i in range(B)], 0)
cond = m._prepare_cond(clips, sr, 48000)
res = []
for rep in range(3):
    mel = m.sample(cond=cond, time_steps=1, noise=noise, decode_to_audio=False).clone()
    wav = m.flowhigh.vocoder.forward(mel).clone()
    out = m.postproc(wav, cond, cond.size(-1)).clone()
    res.append((mel, wav, out))
for rep in (1, 2):
    print(cfgname, B, method, 'rep', rep, 'mel', (res[rep][0]-res[0][0]).abs().max().item(), 'wav', (res[rep][1]-res[0][1]).abs().max().item(), 'out', (res[rep][2]-res[0][2]).abs().max().item())
# vocoder alone, same mel, repeated
mel = res[0][0]
w = [m.flowhigh.vocoder.forward(mel).clone() for _ in range(4)]
print('vocoder-only repeat diffs', [(w[i]-w[0]).abs().max().item() for i in range(1,4)])
