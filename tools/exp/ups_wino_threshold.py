"""Round 6: from which input-channel count on should the transposed convs of a bf16 x 6 model run as Winograd F(4,3) phase groups
(bf16 x 6) instead of the direct fp32 kernel?  (fp32 forms: 768, planner.WINO_UPS_MIN_CIN.)  Times the vocoder's conv launches
per threshold.   python tools/exp/ups_wino_threshold.py   (GPU box)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from flowhigh_amd import synth               # noqa: E402
from flowhigh_amd import vocoder as V        # noqa: E402

cfg = synth.SYNTH_CFG
sd = synth.make_vocoder_state_dict(cfg, 0)
mel = torch.randn(1, 1000, cfg["num_mels"], generator=torch.Generator().manual_seed(0)) * 2.0 - 3.0
ref = None
for thr in (768, 384, 192, 96, 48):
    V.wino_ups_min_cin = lambda form, thr=thr: thr
    voc = V.Vocoder(cfg, sd, "cuda:0")
    p = voc.plan(1, 1000)
    p["mel_in"].copy_(mel.transpose(1, 2).to("cuda:0"))
    for _ in range(3):
        voc.run(p)
    torch.cuda.synchronize()
    launches = p["conv_launches"]
    acc = [0.0] * len(launches)
    R = 10
    for _ in range(R):
        voc.conv_timing = []
        voc.run(p)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(voc.conv_timing):
            acc[i] += a.elapsed_time(b) * 1e3 / R
    voc.conv_timing = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        voc.run(p)
    e1.record()
    torch.cuda.synchronize()
    wav = voc.forward(mel.to("cuda:0")).clone()
    if ref is None:
        ref = wav
    ups = [(f, round(t, 1)) for (f, _, _), t in zip(launches, acc) if f.startswith(("wino43", "direct"))]
    print(f"threshold {thr:4d}: vocoder {e0.elapsed_time(e1) / 20:.3f} ms, conv launches {sum(acc) / 1e3:.3f} ms, max |wav - wav(768)| {float((wav - ref).abs().max()):.2e}")
    print("    ", ups)
