"""Winograd F(5,4) kernel (conv_wino54.hip) against the F(4,3) kernel (conv_wino.hip) on the residual-stack launch shapes
(3 groups k = 11 / 7 / 3, bias + one residual), and both against float64 on a small case.
python tools/wino54_bench.py [batch]"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
DEV = torch.device('cuda:0')
KS = [11, 7, 3]
st = hip.stream()
lib = hip.lib()


def bench(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def check(c, L, d, k, pm, nres, cfg54):
    g = torch.Generator().manual_seed(c + L + d + k)
    x = torch.randn(2, c, L, generator=g)
    w = torch.randn(c, c, k, generator=g) / (c * k) ** 0.5
    b = torch.randn(c, generator=g)
    res = [torch.randn(2, c, L, generator=g) for _ in range(nres)]
    ref = (F.conv1d(x.double(), w.double(), b.double(), dilation=d, padding=(k - 1) // 2 * d) + sum(r.double() for r in res)) * 0.5
    lay = (lambda t: V.to_phase_major(t, d)) if pm else (lambda t: t)
    xd = lay(x).to(DEV)
    rd = [lay(r).to(DEV) for r in res]
    bm = lib.fh_wino54_tile_m(cfg54)
    cpad = -(-c // bm) * bm
    out = torch.full_like(xd, float("nan"))
    u = V.pack_wino54_weight(w, cpad).to(DEV)
    seg = V.make_wino_seg(xd, u, c, k)
    seg.ngrp = -(-k // 4)
    bd = b.to(DEV)
    grp = V.make_wino_group([seg], bd, rd, out, c, cpad, L, scale=0.5)
    desc = hip.to_device_struct_array([grp], DEV)
    hip.check(lib.fh_conv_wino54_f32(desc.data_ptr(), 1, 2, cpad, L, d, int(pm), cfg54, st), "wino54")
    torch.cuda.synchronize()
    got = (V.from_phase_major(out, d, L) if pm else out).cpu().double()
    err = (got - ref).abs().max().item()
    print(f"check C={c} L={L} d={d} k={k} pm={int(pm)} nres={nres} cfg={cfg54}: max err {err:.2e} {'OK' if err < 3e-5 else 'FAIL'}")
    return err < 3e-5


ok = True
for args in ((128, 640, 1, 11, False, 1, 0), (128, 1000, 1, 7, False, 0, 0), (256, 2000, 1, 3, False, 2, 0), (96, 1284, 1, 11, False, 1, 1),
             (64, 644, 1, 7, False, 1, 2), (128, 999, 3, 11, True, 1, 0), (192, 2001, 5, 7, True, 1, 1), (128, 320, 1, 9, False, 0, 0),
             (128, 324, 1, 5, False, 3, 0), (384, 5000, 1, 11, False, 1, 0)):
    ok &= check(*args)
print("ALL OK" if ok else "SOME FAILED")

print(f"{'C':>5} {'L':>7} {'d':>2} {'GFLOP':>8} {'F(4,3) us':>10} {'eff TF/s':>8} {'F(5,4) us':>10} {'eff TF/s':>8} {'ratio':>6}")
for c, L in ((768, 5000), (384, 20000), (192, 60000), (96, 120000)):
    for d in (1, 3):
        pm = d > 1
        pl = d * V.phase_len(L, d) if pm else L
        xs = [torch.randn(B, c, pl, device=DEV) for _ in KS]
        rs = [torch.randn(B, c, pl, device=DEV) for _ in KS]
        outs = [torch.empty(B, c, pl, device=DEV) for _ in KS]
        ws = [torch.randn(c, c, k) * 0.02 for k in KS]
        bs = [torch.randn(c, device=DEV) for _ in KS]
        wcfg, wpad = V.pick_wino_tile(c)
        ud = [V.pack_wino_weight(w, wpad).to(DEV) for w in ws]
        gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [rs[i]], outs[i], c, wpad, L) for i, k in enumerate(KS)]
        wcfg, _ = V.choose_wino_cfg([c // 16 * -(-k // 3) for k in KS], B, wpad, L, d, default=wcfg)
        dw = hip.to_device_struct_array(gw, DEV)
        cfg54 = 0 if c % 128 == 0 else 1 if c % 96 == 0 else 2
        bm = lib.fh_wino54_tile_m(cfg54)
        cpad = -(-c // bm) * bm
        u5 = [V.pack_wino54_weight(w, cpad).to(DEV) for w in ws]
        g5 = []
        for i, k in enumerate(KS):
            seg = V.make_wino_seg(xs[i], u5[i], c, k)
            seg.ngrp = -(-k // 4)
            g5.append(V.make_wino_group([seg], bs[i], [rs[i]], outs[i], c, cpad, L))
        d5 = hip.to_device_struct_array(g5, DEV)
        t43 = bench(lambda: hip.check(lib.fh_conv_wino_f32(dw.data_ptr(), 3, B, wpad, L, d, int(pm), wcfg, st)))
        t54 = bench(lambda: hip.check(lib.fh_conv_wino54_f32(d5.data_ptr(), 3, B, cpad, L, d, int(pm), cfg54, st)))
        fl = 2.0 * c * c * sum(KS) * L * B
        print(f"{c:5d} {L:7d} {d:2d} {fl/1e9:8.2f} {t43:10.1f} {fl/t43/1e6:8.1f} {t54:10.1f} {fl/t54/1e6:8.1f} {t43/t54:6.2f}")
