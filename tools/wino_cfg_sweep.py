"""Time one residual-stack launch shape under every compatible Winograd tile (cfg 0: 64x512, 1: 96x256,
4: 64x256, 5: 32x256) - data for the plan-time tile choice.  python tools/wino_cfg_sweep.py"""
import sys
import torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V

DEV = torch.device('cuda:0')
KS = [11, 7, 3]
st = hip.stream()


def time_launch(dw, ng, B, wpad, L, d, cfg, reps=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), ng, B, wpad, L, d, 0, cfg, st))
    ev[0].record()
    for _ in range(reps):
        hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), ng, B, wpad, L, d, 0, cfg, st))
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


def shape(c, L, closing, B=1):
    xs = [torch.randn(B, c, L, device=DEV) for _ in KS]
    outs = [torch.empty(B, c, L, device=DEV) for _ in KS]
    bs = [torch.randn(c, device=DEV) for _ in KS]
    wpad = -(-c // 192) * 192                 # divisible by 64 and 96: same packed weights for every tile
    ud = [V.pack_wino_weight(torch.randn(c, c, k) * 0.02, wpad).to(DEV) for k in KS]
    if closing:
        gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k) for i, k in enumerate(KS)], bs[0], [], outs[0],
                                c, wpad, L)]
    else:
        gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [], outs[i], c, wpad, L)
              for i, k in enumerate(KS)]
    dw = hip.to_device_struct_array(gw, DEV)
    row = []
    ref = None
    for cfg, (bm, bt) in ((0, (64, 512)), (1, (96, 256)), (4, (64, 256)), (5, (32, 256)), (6, (128, 256))):
        if wpad % bm:
            continue
        blocks = B * len(gw) * (wpad // bm) * -(-L // bt)
        t = time_launch(dw, len(gw), B, wpad, L, 1, cfg)
        if ref is None:
            ref = [o.clone() for o in outs]
        err = max(float((o - r).abs().max()) for o, r in zip(outs, ref))
        row.append(f"cfg{cfg} {blocks:5d} blk {t:7.1f} us" + (f" (diff {err:.1e})" if cfg else ""))
    print(f"c={c:4d} L={L:6d} {'closing' if closing else 'stack  '} | " + " | ".join(row), flush=True)


for secs in (10, 5):
    n = secs * 100
    for c, up in ((768, 5), (384, 20), (192, 60)):
        for closing in (False, True):
            shape(c, n * up, closing)
shape(768, 5000, False, 8)
shape(384, 20000, False, 8)
