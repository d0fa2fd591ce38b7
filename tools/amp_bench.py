"""Time the narrow-stage kernel (fh_amp_actconv_f32) on a stage-sized launch next to the unfused launches it replaces.
    python tools/amp_bench.py [C=48] [L=240000] [d=1] [reps=50]
Three groups (k = 11 / 7 / 3) with bias and one residual, random data."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowhigh_amd import hip, synth          # noqa: E402
from flowhigh_amd import vocoder as V        # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 48
L = int(sys.argv[2]) if len(sys.argv) > 2 else 240000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
ks = (11, 7, 3)
filt = synth.kaiser_sinc_filter().flatten().tolist()
p = dict(alpha=torch.rand(C, generator=g).add(0.5).to(dev), inv_beta=torch.rand(C, generator=g).add(0.5).to(dev), up=filt, down=filt)
xs = [torch.randn(1, C, L, generator=g).to(dev) for _ in ks]
rs = [torch.randn(1, C, L, generator=g).to(dev) for _ in ks]
outs = [torch.empty(1, C, L, device=dev) for _ in ks]
ws = [torch.randn(C, C, k, generator=g) * (C * k) ** -0.5 for k in ks]
bias = torch.zeros(C, device=dev)
us = [V.pack_amp_weight(w, C).to(dev) for w in ws]


def timed(fn, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


lib, st = hip.lib(), hip.stream()
for act in (False,):          # (the form with the activation inside the launch left the library with ABI 4)
    groups = [V.make_amp_group([V.make_amp_seg(xs[i], us[i], k)], bias, [rs[i]], outs[i], L) for i, k in enumerate(ks)]
    tiles = V.amp_tile_list([L] * 3, 1, d).to(dev)
    total = tiles.shape[0]
    desc = hip.to_device_struct_array(groups, dev)
    flags = int(L % 4 == 0) | 2
    us_ = timed(lambda: hip.check(lib.fh_amp_actconv_f32(desc.data_ptr(), 3, tiles.data_ptr(), total, C, d, 5, flags, st), "amp"))
    print(f"C={C} L={L} d={d} fused act={act}: {us_:.1f} us  ({total} blocks)")
# the unfused pair: activation launch + the conv launch of the model at this width
ys = [torch.empty(1, C, d * V.phase_len(L, d) if d > 1 else L, device=dev) for _ in ks]
ga = hip.to_device_struct_array([V.make_act_group(xs[i], ys[i], p) for i in range(3)], dev)
t_act = timed(lambda: hip.check(lib.fh_act1d_grouped_pm_f32(ga.data_ptr(), 3, 1, C, L, 1, d, st), "act"))
print(f"  activation launch alone: {t_act:.1f} us")
if V.use_wino54(C):
    wcfg, wpad = V.pick_wino54_tile(C)
    uw = [V.pack_wino54_weight(w, wpad).to(dev) for w in ws]
    pm = d > 1
    o2 = [torch.empty_like(y) for y in ys]
    r2 = [torch.randn_like(y) for y in ys]
    gw = hip.to_device_struct_array([V.make_wino_group([V.make_wino_seg(ys[i], uw[i], C, k, taps=4)], bias, [r2[i]], o2[i], C, wpad, L)
                                     for i, k in enumerate(ks)], dev)
    t_conv = timed(lambda: hip.check(lib.fh_conv_wino54_f32(gw.data_ptr(), 3, 1, wpad, L, d, int(pm), wcfg & 15, st), "w54"))
    print(f"  F(5,4) conv launch alone ({wpad} rows): {t_conv:.1f} us")
else:
    tcfg, bm, cpad = V.pick_tile_cfg(C)
    ck = V.pick_ck(C)
    wd = [V.pack_conv_weight(w, cpad, ck).to(dev) for w in ws]
    gc = hip.to_device_struct_array([V.make_conv_group([V.make_conv_seg(xs[i], wd[i], C, [(t - (k - 1) // 2) * d for t in range(k)])],
                                                       bias, [rs[i]], outs[i], C, cpad, L, L, L) for i, k in enumerate(ks)], dev)
    t_conv = timed(lambda: hip.check(lib.fh_conv_grouped_f32(gc.data_ptr(), 3, 1, cpad, L, tcfg, ck, st), "conv"))
    print(f"  direct conv launch alone: {t_conv:.1f} us")
