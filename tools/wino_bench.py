"""Winograd F(4,3) conv vs the direct implicit-GEMM conv on the residual-stack launch shapes
(3 groups k = 11 / 7 / 3 per launch).  python tools/wino_bench.py [batch]"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
DEV = torch.device('cuda:0')
KS = [11, 7, 3]

def bench(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

print(f"{'C':>5} {'L':>7} {'d':>2} {'GFLOP':>8} {'direct us':>10} {'TF/s':>6} {'wino us':>9} {'eff TF/s':>8} {'speedup':>7}")
for c, L in ((768, 5000), (384, 20000), (192, 60000), (96, 120000), (48, 240000)):
    for d in (1, 3, 5):
        xs = [torch.randn(B, c, L, device=DEV) for _ in KS]
        outs = [torch.empty(B, c, L, device=DEV) for _ in KS]
        ws = [torch.randn(c, c, k) * 0.02 for k in KS]
        bs = [torch.randn(c, device=DEV) for _ in KS]
        tcfg, _, cpad = V.pick_tile_cfg(c)
        ck = V.pick_ck(c)
        wd = [V.pack_conv_weight(w, cpad, ck).to(DEV) for w in ws]
        gd = [V.make_conv_group([V.make_conv_seg(xs[i], wd[i], c, [(t - (k - 1) // 2) * d for t in range(k)])],
                                bs[i], [], outs[i], c, cpad, L, L, L) for i, k in enumerate(KS)]
        dd = hip.to_device_struct_array(gd, DEV)
        wcfg, wpad = V.pick_wino_tile(c)
        import os
        if os.environ.get("WCFG"):
            wcfg = int(os.environ["WCFG"])
        ud = [V.pack_wino_weight(w, wpad).to(DEV) for w in ws]
        gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [], outs[i], c, wpad, L)
              for i, k in enumerate(KS)]
        dw = hip.to_device_struct_array(gw, DEV)
        st = hip.stream()
        t_d = bench(lambda: hip.check(hip.lib().fh_conv_grouped_f32(dd.data_ptr(), 3, B, cpad, L, tcfg, ck, st)))
        t_w = bench(lambda: hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), 3, B, wpad, L, d, 0, wcfg, st)))
        t_p = float('nan')
        if d > 1:      # same launch on phase-major tensors (timing only: the buffers are reinterpreted)
            pl = d * V.phase_len(L, d)
            xs2 = [torch.randn(B, c, pl, device=DEV) for _ in KS]
            outs2 = [torch.empty(B, c, pl, device=DEV) for _ in KS]
            gp = [V.make_wino_group([V.make_wino_seg(xs2[i], ud[i], c, k)], bs[i], [], outs2[i], c, wpad, L)
                  for i, k in enumerate(KS)]
            dp = hip.to_device_struct_array(gp, DEV)
            t_p = bench(lambda: hip.check(hip.lib().fh_conv_wino_f32(dp.data_ptr(), 3, B, wpad, L, d, 1, wcfg, st)))
        fl = 2.0 * c * c * sum(KS) * L * B
        print(f"{c:5d} {L:7d} {d:2d} {fl/1e9:8.2f} {t_d:10.1f} {fl/t_d/1e6:6.1f} {t_w:9.1f} {fl/t_w/1e6:8.1f} {t_d/t_w:7.2f}  pm {t_p:7.1f}")
