"""Time one Winograd launch shape (us): 3 groups k = 11 / 7 / 3.  python tools/wino_time.py C L d [tile_cfg ...] [bf]
bf: also the three-piece bf16 form (tile_cfg | FH_WINO_BF16X6)."""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V
args = [a for a in sys.argv[1:] if a != "bf"]
BF = "bf" in sys.argv[1:]
c, L, d, B = int(args[0]), int(args[1]), int(args[2]), 1
DEV = torch.device('cuda:0'); KS = [11, 7, 3]
xs = [torch.randn(B, c, L, device=DEV) for _ in KS]
outs = [torch.empty(B, c, L, device=DEV) for _ in KS]
ws = [torch.randn(c, c, k) * 0.02 for k in KS]
bs = [torch.randn(c, device=DEV) for _ in KS]
wcfg0, wpad = V.pick_wino_tile(c)
cfgs = [int(a) for a in args[3:]] or [wcfg0]
st = hip.stream()
fl = 2.0 * c * c * 1.5 * sum(-(-k // 3) for k in KS) * L * B          # executed (Winograd) FLOPs
for mode in ([0, V.WINO_BF16X6] if BF else [0]):
    ud = [V.pack_wino_weight(w, wpad) for w in ws]
    ud = [(V.split_bf3(u) if mode else u).to(DEV) for u in ud]
    gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [], outs[i], c, wpad, L) for i, k in enumerate(KS)]
    dw = hip.to_device_struct_array(gw, DEV)
    for cfg in cfgs:
        if wpad % hip.lib().fh_wino_tile_m(cfg):
            continue
        run = lambda: hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), 3, B, wpad, L, d, 0, cfg | mode, st))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"C={c} L={L} d={d} cfg={cfg} {'bf16x6' if mode else 'fp32  '}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s executed (fp32-equivalent)")
