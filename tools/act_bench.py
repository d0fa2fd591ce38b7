"""Activation1d launch timing, plain vs phase-major layouts.  python tools/act_bench.py"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, synth, vocoder as V
DEV = torch.device('cuda:0')
import os
if os.environ.get('FH_ACT_BLOCKS', 'auto') != 'auto':          # (no Vocoder here: force the occupancy cap by hand)
    hip.check(hip.lib().fh_act_set_blocks_per_cu(int(os.environ['FH_ACT_BLOCKS'])))
filt = synth.kaiser_sinc_filter().flatten().tolist()
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for c, L in ((768, 5000), (384, 20000), (192, 60000), (96, 120000), (48, 240000)):
    row = []
    for din, dout in ((1, 1), (1, 3), (3, 1), (1, 5), (5, 1)):
        n = max(L, max(d * V.phase_len(L, d) for d in (din, dout)))
        xs = [torch.randn(1, c, n, device=DEV) for _ in range(3)]
        ys = [torch.empty(1, c, n, device=DEV) for _ in range(3)]
        p = dict(alpha=torch.rand(c, device=DEV) + 0.5, inv_beta=torch.rand(c, device=DEV) + 0.5, up=filt, down=filt)
        g = hip.to_device_struct_array([V.make_act_group(xs[i], ys[i], p) for i in range(3)], DEV)
        st = hip.stream()
        t = bench(lambda: hip.check(hip.lib().fh_act1d_grouped_pm_f32(g.data_ptr(), 3, 1, c, L, din, dout, st)))
        row.append(f"{din}->{dout}: {t:6.1f} us {3 * c * L * 8 / t / 1e6:5.2f} TB/s")
    print(f"C={c:4d} L={L:6d}  " + "  ".join(row))
