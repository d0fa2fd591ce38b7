"""What makes a Winograd launch slower when it follows an activation launch (tools/box_probe.sh: 2.11 instead of 2.38 GHz on
some boxes)?  The same conv launch, 300 times, each preceded by one 'intermission': nothing, the real activation launch, a
device copy of the same bytes, a read-only reduction of the same bytes, an idle gap of the same length (one spinning
block), the activation launch on a tenth of the data, a register-only vector-ALU kernel of the same length.
python tools/clock_dip_probe.py"""
import sys, time, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, synth, vocoder as V
DEV = torch.device('cuda:0'); KS = [11, 7, 3]; st = hip.stream()
c, L = 192, 60000
xs = [torch.randn(1, c, L, device=DEV) for _ in KS]; outs = [torch.empty(1, c, L, device=DEV) for _ in KS]
bs = [torch.randn(c, device=DEV) for _ in KS]
wcfg, wpad = V.pick_wino_tile(c)
ud = [V.pack_wino_weight(torch.randn(c, c, k) * 0.02, wpad).to(DEV) for k in KS]
gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [], outs[i], c, wpad, L) for i, k in enumerate(KS)]
dw = hip.to_device_struct_array(gw, DEV, 128)
conv = lambda: hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), 3, 1, wpad, L, 1, 0, wcfg, st))
filt = synth.kaiser_sinc_filter().flatten().tolist()
p = dict(alpha=torch.rand(c, device=DEV) + 0.5, inv_beta=torch.rand(c, device=DEV) + 0.5, up=filt, down=filt)
ya = [torch.empty(1, c, L, device=DEV) for _ in KS]
xa = [torch.randn(1, c, L, device=DEV) for _ in KS]
ga = hip.to_device_struct_array([V.make_act_group(xa[i], ya[i], p) for i in range(3)], DEV)
act = lambda: hip.check(hip.lib().fh_act1d_grouped_pm_f32(ga.data_ptr(), 3, 1, c, L, 1, 1, st))
ga_small = hip.to_device_struct_array([V.make_act_group(xa[i], ya[i], p) for i in range(3)], DEV)
act_small = lambda: hip.check(hip.lib().fh_act1d_grouped_pm_f32(ga_small.data_ptr(), 3, 1, c, L // 10, 1, 1, st))
big_in = torch.randn(3 * c * L, device=DEV); big_out = torch.empty_like(big_in)
copy = lambda: big_out.copy_(big_in)
reduce_ = lambda: torch.sum(big_in)
spin_cycles = int(60e-6 * 100e6 * 24)          # torch.cuda._sleep counts shader-ish cycles; calibrated below
def calibrate():
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
    return 1000000 / (e0.elapsed_time(e1) * 1e3)     # cycles per us
cpu = calibrate()
idle = lambda: torch.cuda._sleep(int(60 * cpu))
# register-only vector ALU work, chip-wide, ~60 us: an elementwise chain on a small (L2-resident) tensor
small = torch.randn(256 * 1024, device=DEV)
def alu():
    t = small
    for _ in range(6):
        t = torch.sin(t)
    return t
def time_of(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
import os
if os.environ.get("ONLY_ACT"):               # (for runs against ablation / variant builds of the activation kernel: FH_LIB_PATH)
    cases = [("nothing", None), ("activation launch [" + os.environ["ONLY_ACT"] + "]", act)]
else:
  cases = [("nothing", None), ("activation launch", act), ("device copy, same bytes", copy), ("read-only sum, same bytes", reduce_),
           ("idle gap (one spinning block)", idle), ("activation on a tenth of the rows' length", act_small), ("small elementwise chain", alu)]
for name, fn in cases:
    dur = time_of(fn) if fn else 0.0
    time.sleep(0.3)
    for _ in range(100):                      # ramp
        if fn: fn()
        conv()
    ce = []
    for _ in range(300):
        if fn: fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); conv(); b.record(); ce.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ce)
    print(f"{name:44s} ({dur:6.1f} us each): conv launch median {t[len(t) // 2]:6.1f} us, p10 {t[30]:6.1f}, p90 {t[270]:6.1f}")

# how long does the lower clock last?  one activation launch, then four conv launches back to back
if not os.environ.get("ONLY_ACT"):
    time.sleep(0.3)
    for _ in range(60):
        act()
        for _ in range(4): conv()
    rows = []
    for _ in range(100):
        act()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        for k in range(4):
            conv(); ev[k + 1].record()
        rows.append(ev)
    torch.cuda.synchronize()
    med = lambda v: sorted(v)[len(v) // 2]
    print("one activation launch, then four conv launches: median us of the 1st .. 4th:",
          " ".join(f"{med([r[k].elapsed_time(r[k + 1]) * 1e3 for r in rows]):6.1f}" for k in range(4)))
