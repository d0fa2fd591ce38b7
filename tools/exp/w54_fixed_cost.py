"""Fixed cost per block of the F(5,4) kernel: the merged residual-stack launch (3 groups k = 11 / 7 / 3, 96 output channels,
120 000 samples = 1 125 blocks) with the input channel count swept: time = a + b cin; a / (blocks per CU) is what a block costs
before and after its K loop.  python tools/exp/w54_fixed_cost.py [cout] [len]"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V

DEV = torch.device('cuda:0')
KS = [11, 7, 3]
st = hip.stream()
lib = hip.lib()
cout = int(sys.argv[1]) if len(sys.argv) > 1 else 96
L = int(sys.argv[2]) if len(sys.argv) > 2 else 120000


def bench(fn, reps=30):
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


cfg54 = 0 if cout % 128 == 0 else 1 if cout % 96 == 0 else 3 if cout % 48 == 0 else 2
bm = lib.fh_wino54_tile_m(cfg54)
cpad = -(-cout // bm) * bm
print(f"cout {cout} len {L} tile rows {bm}: {3 * (cpad // bm) * lib.fh_wino54_n_tiles(L, 1, 0)} blocks")
for nres in (0, 1):
    pts = []
    for cin in (16, 32, 48, 96, 192, 384):
        xs = [torch.randn(1, cin, L, device=DEV) for _ in KS]
        rs = [torch.randn(1, cout, L, device=DEV) for _ in KS]
        outs = [torch.empty(1, cout, L, device=DEV) for _ in KS]
        bs = [torch.randn(cout, device=DEV) for _ in KS]
        us = [V.pack_wino54_weight(torch.randn(cout, cin, k) * 0.02, cpad).to(DEV) for k in KS]
        gs = []
        for i, k in enumerate(KS):
            seg = V.make_wino_seg(xs[i], us[i], cin, k)
            seg.ngrp = -(-k // 4)
            gs.append(V.make_wino_group([seg], bs[i], [rs[i]] if nres else [], outs[i], cout, cpad, L))
        d = hip.to_device_struct_array(gs, DEV)
        t = bench(lambda: hip.check(lib.fh_conv_wino54_f32(d.data_ptr(), 3, 1, cpad, L, 1, 0, cfg54, st), "w54"))
        pts.append((cin, t))
        print(f"  nres {nres} cin {cin:4d}: {t:8.1f} us")
    n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
    sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
    print(f"nres {nres}: time = {a:.1f} us + {b:.3f} us per input channel  (per 96 channels: {96 * b:.1f} us)")
