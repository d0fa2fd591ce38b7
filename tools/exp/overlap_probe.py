"""Round 6: can two launches with different bottlenecks overlap on this chip?  A narrow-stage conv launch (direct bf16 x 6, C = 24,
L = 480 000, three groups: matrix work + 345-415 MB) and an Activation1d launch of the same size (276 MB, vector work) run
(a) each alone, (b) one after the other on one stream, (c) on two streams at once.  If (c) is close to max(a) the chip overlaps
them and the step's launches could be scheduled against each other; if (c) is close to (b) = sum(a), time adds over what the
chip does, whoever issues it.      python tools/exp/overlap_probe.py [C=24] [L=480000]      (GPU box)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from flowhigh_amd import hip, synth          # noqa: E402
from flowhigh_amd import vocoder as V        # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = int(sys.argv[2]) if len(sys.argv) > 2 else 480000
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
ks = (11, 7, 3)
xs = [torch.randn(1, C, L, generator=g).to(dev) for _ in ks]
rs = [torch.randn(1, C, L, generator=g).to(dev) for _ in ks]
outs = [torch.empty(1, C, L, device=dev) for _ in ks]
bias = torch.zeros(C, device=dev)
us = [V.pack_narrow_bf_weight(torch.randn(C, C, k, generator=g) * (C * k) ** -0.5, C).to(dev) for k in ks]
groups = [V.make_amp_group([V.make_amp_seg(xs[i], us[i], k, direct=True)], bias, [rs[i]], outs[i], L, direct=True) for i, k in enumerate(ks)]
tiles = V.amp_tile_list([L] * 3, 1, 1, direct=True).to(dev)
desc = hip.to_device_struct_array(groups, dev)
lib = hip.lib()
# the activation launch works on its own tensors (no dependence between the two launches)
filt = synth.kaiser_sinc_filter().flatten().tolist()
p = dict(alpha=torch.rand(C, generator=g).add(0.5).to(dev), inv_beta=torch.rand(C, generator=g).add(0.5).to(dev), up=filt, down=filt)
ax = [torch.randn(1, C, L, generator=g).to(dev) for _ in ks]
ay = [torch.empty(1, C, L, device=dev) for _ in ks]
ga = hip.to_device_struct_array([V.make_act_group(ax[i], ay[i], p) for i in range(3)], dev)


def conv(st):
    hip.check(lib.fh_narrow_conv_bf16x6_f32(desc.data_ptr(), 3, tiles.data_ptr(), tiles.shape[0], C, 1, int(L % 4 == 0), st), "narrow")


def act(st):
    hip.check(lib.fh_act1d_grouped_pm_f32(ga.data_ptr(), 3, 1, C, L, 1, 1, st), "act")


s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
REPS = 40


def timed(fn):
    for _ in range(200):         # ~30-60 ms of load: the clock ramps for ~20 ms after idle
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


timed(lambda: conv(s0.cuda_stream))       # (the first ~100 ms of a process run these launches 2-3 x slower: not timed)
a = timed(lambda: conv(s1.cuda_stream))
b = timed(lambda: act(s1.cuda_stream))
c = timed(lambda: (conv(s1.cuda_stream), act(s1.cuda_stream)))
d = timed(lambda: (conv(s1.cuda_stream), act(s2.cuda_stream)))
print(f"C = {C}, L = {L}: conv alone {a:.1f} us, activation alone {b:.1f} us, one stream {c:.1f} us (sum {a + b:.1f}), two streams {d:.1f} us "
      f"(max {max(a, b):.1f}); overlap gain {100 * (c - d) / c:.0f} % of the serial pair")
