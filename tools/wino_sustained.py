"""Does a Winograd launch get faster under SUSTAINED load (clock ramp)?  The same launch back to back for ~0.4 s, time per
group of 40 launches; then the same interleaved with an activation launch (the model's real alternation).
python tools/wino_sustained.py [C] [L]"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, synth, vocoder as V
DEV = torch.device('cuda:0'); KS = [11, 7, 3]; st = hip.stream()
c = int(sys.argv[1]) if len(sys.argv) > 1 else 192
L = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
xs = [torch.randn(1, c, L, device=DEV) for _ in KS]; outs = [torch.empty(1, c, L, device=DEV) for _ in KS]
bs = [torch.randn(c, device=DEV) for _ in KS]
wcfg, wpad = V.pick_wino_tile(c)
ud = [V.pack_wino_weight(torch.randn(c, c, k) * 0.02, wpad).to(DEV) for k in KS]
gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [], outs[i], c, wpad, L) for i, k in enumerate(KS)]
dw = hip.to_device_struct_array(gw, DEV, 128)
conv = lambda: hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), 3, 1, wpad, L, 1, 0, wcfg, st))
filt = synth.kaiser_sinc_filter().flatten().tolist()
p = dict(alpha=torch.rand(c, device=DEV) + 0.5, inv_beta=torch.rand(c, device=DEV) + 0.5, up=filt, down=filt)
ga = hip.to_device_struct_array([V.make_act_group(outs[i], xs[i], p) for i in range(3)], DEV)
act = lambda: hip.check(hip.lib().fh_act1d_grouped_pm_f32(ga.data_ptr(), 3, 1, c, L, 1, 1, st))
torch.cuda.synchronize()
import time; time.sleep(0.5)                      # start from an idle chip
for name, body in (("conv only", lambda: conv()), ("act + conv alternating", lambda: (act(), conv()))):
    evs = []
    for g in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ce = []
        e0.record()
        for _ in range(40):
            if name != "conv only":
                act()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); conv(); b.record(); ce.append((a, b))
        e1.record()
        evs.append((e0, e1, ce))
    torch.cuda.synchronize()
    print(name, "C", c, "L", L, "| us per conv launch, groups of 40 in time order:",
          " ".join(f"{sum(a.elapsed_time(b) for a, b in ce) / 40 * 1e3:.0f}" for _, _, ce in evs))
    time.sleep(0.5)
