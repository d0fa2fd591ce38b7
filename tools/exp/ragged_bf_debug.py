"""Round 6: which run of the bf16 x 6 vocoder is unstable -- the clip alone or the ragged batch -- and where."""
import os, sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import synth, vocoder as V
DEV = torch.device("cuda:0")
cfg = synth.SYNTH_CFG
frames = [50, 333, 50, 77, 201, 3]
sd = synth.make_vocoder_state_dict(cfg, seed=1)
g = lambda n, seed: (torch.randn(n, 256, generator=torch.Generator().manual_seed(seed)) * 2.0 - 3.0).to(DEV)
mels = [g(n, 180 + i) for i, n in enumerate(frames)]
for form in ("bf16x6", "winograd"):
    voc = V.Vocoder(cfg, sd, DEV, conv_form=form)
    a1 = [voc.forward(m[None]).clone() for m in mels]
    a2 = [voc.forward(m[None]).clone() for m in mels]
    r1 = [t.clone() for t in voc.forward_ragged(mels)]
    r2 = [t.clone() for t in voc.forward_ragged(mels)]
    for i, n in enumerate(frames):
        d = lambda x, y: (x - y).abs()
        bad = (d(a1[i], r1[i]) > 0).nonzero()
        print(form, f"clip {i} ({n} frames): alone vs alone {d(a1[i], a2[i]).max().item():.2e}  ragged vs ragged {d(r1[i], r2[i]).max().item():.2e}  "
              f"alone vs ragged {d(a1[i], r1[i]).max().item():.2e}" + (f"  first/last differing sample {bad[0, 1].item()} / {bad[-1, 1].item()} of {a1[i].shape[1]}" if bad.numel() else ""))
