"""Time one Winograd launch shape (us).  python tools/wino_time.py C L d"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V
c, L, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 1
DEV = torch.device('cuda:0'); KS = [11, 7, 3]
xs = [torch.randn(B, c, L, device=DEV) for _ in KS]
outs = [torch.empty(B, c, L, device=DEV) for _ in KS]
ws = [torch.randn(c, c, k) * 0.02 for k in KS]
bs = [torch.randn(c, device=DEV) for _ in KS]
wcfg, wpad = V.pick_wino_tile(c)
ud = [V.pack_wino_weight(w, wpad).to(DEV) for w in ws]
gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [], outs[i], c, wpad, L) for i, k in enumerate(KS)]
dw = hip.to_device_struct_array(gw, DEV)
st = hip.stream()
run = lambda: hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), 3, B, wpad, L, d, 0, wcfg, st))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 100:.1f} us")
