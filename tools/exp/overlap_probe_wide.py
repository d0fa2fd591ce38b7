"""Round 6: the two-stream question of overlap_probe.py for a WIDE-stage launch: an F(5,4) bf16 x 6 conv launch (three groups
k = 11 / 7 / 3, bias + residual; one 8-wave block per CU) and an Activation1d launch of the same size on its own tensors: alone,
on one stream, on two streams.      python tools/exp/overlap_probe_wide.py [C=192] [L=60000]      (GPU box)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from flowhigh_amd import hip, synth          # noqa: E402
from flowhigh_amd import vocoder as V        # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 192
L = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
ks = (11, 7, 3)
lib = hip.lib()
wcfg, cpad = V.pick_wino54_tile(C, True)
xs = [torch.randn(1, C, L, generator=g).to(dev) for _ in ks]
rs = [torch.randn(1, C, L, generator=g).to(dev) for _ in ks]
outs = [torch.empty(1, C, L, device=dev) for _ in ks]
bs = [torch.randn(C, generator=g).to(dev) for _ in ks]
us = [V.pack_wino54_weight_any(torch.randn(C, C, k, generator=g) * (C * k) ** -0.5, cpad, True).to(dev) for k in ks]
groups = [V.make_wino_group([V.make_wino_seg(xs[i], us[i], C, k, taps=4)], bs[i], [rs[i]], outs[i], C, cpad, L) for i, k in enumerate(ks)]
desc = hip.to_device_struct_array(groups, dev)
tile_cfg = (wcfg & 15) | V.WINO_BF16X6
filt = synth.kaiser_sinc_filter().flatten().tolist()
p = dict(alpha=torch.rand(C, generator=g).add(0.5).to(dev), inv_beta=torch.rand(C, generator=g).add(0.5).to(dev), up=filt, down=filt)
ax = [torch.randn(1, C, L, generator=g).to(dev) for _ in ks]
ay = [torch.empty(1, C, L, device=dev) for _ in ks]
ga = hip.to_device_struct_array([V.make_act_group(ax[i], ay[i], p) for i in range(3)], dev)


def conv(st):
    hip.check(lib.fh_conv_wino54_f32(desc.data_ptr(), 3, 1, cpad, L, 1, 0, tile_cfg, st), "wino54")


def act(st):
    hip.check(lib.fh_act1d_grouped_pm_f32(ga.data_ptr(), 3, 1, C, L, 1, 1, st), "act")


s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
REPS = 40


def timed(fn, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s0)
    s1.wait_stream(s0); s2.wait_stream(s0)
    for _ in range(REPS):
        fn()
    s0.wait_stream(s1); s0.wait_stream(s2)
    e1.record(s0)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


timed(lambda: conv(s0.cuda_stream), warm=300)       # (the first ~100 ms of a process run slower: not timed)
a = timed(lambda: conv(s1.cuda_stream))
b = timed(lambda: act(s1.cuda_stream))
c = timed(lambda: (conv(s1.cuda_stream), act(s1.cuda_stream)))
d = timed(lambda: (conv(s1.cuda_stream), act(s2.cuda_stream)))
# two conv launches of half the groups' work each on two streams against one launch (does a split launch cost anything?)
print(f"C = {C}, L = {L}: conv alone {a:.1f} us, activation alone {b:.1f} us, one stream {c:.1f} us (sum {a + b:.1f}), two streams {d:.1f} us "
      f"(max {max(a, b):.1f}); overlap gain {100 * (c - d) / c:.0f} % of the serial pair")
