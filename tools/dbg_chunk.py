import sys, os, torch
sys.path.insert(0, '.')
from flowhigh_amd import synth, vocoder as V
cfgname, N, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = getattr(synth, cfgname)
sd = synth.make_vocoder_state_dict(cfg, 1)
voc = V.Vocoder(cfg, sd, 'cuda', bf16x6=True)
mel = (torch.randn(1, N, 256, generator=torch.Generator().manual_seed(5)) * 2.0 - 3.0).cuda()
whole = voc.forward(mel).clone()
got = voc.forward_chunked(mel, chunk)
d = (got - whole).abs()
bad = (d > 0).nonzero()
print(cfgname, N, chunk, os.environ.get("FH_WINO_SPLITK"), os.environ.get("FH_WINO_AUTO"), os.environ.get("FH_WINO_NO_VL"),
      "equal" if not len(bad) else f"DIFF max {d.max().item():.2e} at {len(bad)} samples, first {bad[0].tolist()} last {bad[-1].tolist()}")
for k in dict.keys(voc._plans):
    p = dict.get(voc._plans, k)
    print("  plan", k, [(s[7] if s[0] == 'wino' else s[5]) for s in p["steps"] if s[0] in ("wino", "conv")][:20])
