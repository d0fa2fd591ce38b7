"""One residual-stack launch shape, Winograd and direct, a few repetitions (for rocprofv3 --pmc)."""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V
c, L, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 1
which = sys.argv[4] if len(sys.argv) > 4 else "both"
DEV = torch.device('cuda:0'); KS = [11, 7, 3]
xs = [torch.randn(B, c, L, device=DEV) for _ in KS]
outs = [torch.empty(B, c, L, device=DEV) for _ in KS]
ws = [torch.randn(c, c, k) * 0.02 for k in KS]
bs = [torch.randn(c, device=DEV) for _ in KS]
tcfg, _, cpad = V.pick_tile_cfg(c); ck = V.pick_ck(c)
wd = [V.pack_conv_weight(w, cpad, ck).to(DEV) for w in ws]
gd = [V.make_conv_group([V.make_conv_seg(xs[i], wd[i], c, [(t - (k - 1) // 2) * d for t in range(k)])],
                        bs[i], [], outs[i], c, cpad, L, L, L) for i, k in enumerate(KS)]
dd = hip.to_device_struct_array(gd, DEV)
wcfg, wpad = V.pick_wino_tile(c)
ud = [V.pack_wino_weight(w, wpad).to(DEV) for w in ws]
gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [], outs[i], c, wpad, L) for i, k in enumerate(KS)]
dw = hip.to_device_struct_array(gw, DEV)
st = hip.stream()
for _ in range(5):
    if which in ("both", "direct"):
        hip.check(hip.lib().fh_conv_grouped_f32(dd.data_ptr(), 3, B, cpad, L, tcfg, ck, st))
    if which in ("both", "wino"):
        hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), 3, B, wpad, L, d, 0, wcfg, st))
torch.cuda.synchronize()
