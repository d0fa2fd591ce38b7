import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V
DEV = torch.device('cuda:0'); st = hip.stream(); lib = hip.lib()
def run(c, L, k, w, x, cfg54=0, b=None):
    bm = lib.fh_wino54_tile_m(cfg54); cpad = -(-c // bm) * bm
    xd = x.to(DEV); out = torch.full_like(xd, float("nan"))
    u = V.pack_wino54_weight(w, cpad).to(DEV)
    seg = V.make_wino_seg(xd, u, c, k); seg.ngrp = -(-k // 4)
    grp = V.make_wino_group([seg], b.to(DEV) if b is not None else None, [], out, c, cpad, L)
    desc = hip.to_device_struct_array([grp], DEV)
    hip.check(lib.fh_conv_wino54_f32(desc.data_ptr(), 1, x.shape[0], cpad, L, 1, 0, cfg54, st), "wino54")
    torch.cuda.synchronize()
    return out.cpu()
c, L = 128, 320
x = torch.randn(1, c, L)
for k, tap in ((3, 1), (3, 0), (3, 2), (7, 3), (7, 0), (7, 6)):
    w = torch.zeros(c, c, k); w[torch.arange(c), torch.arange(c), tap] = 1.0
    ref = F.conv1d(x, w, padding=(k - 1) // 2)
    out = run(c, L, k, w, x)
    d = (out - ref).abs()
    print(f"k={k} tap={tap}: max err {d.max():.3e}; bad rows {int((d.amax(2) > 1e-4).sum())}/{c}; bad cols {int((d.amax(1) > 1e-4).sum())}/{L}; first bad col {int((d.amax(1)[0] > 1e-4).nonzero()[0]) if (d.amax(1)[0] > 1e-4).any() else -1}")
    if d.max() > 1e-4:
        r = int(d.amax(2)[0].argmax()); print("   row", r, "out", out[0, r, :12].tolist(), "\n   ref", ref[0, r, :12].tolist())
# channel mixing: w[co, ci] random, single tap
w = torch.zeros(c, c, 3); w[:, :, 1] = torch.randn(c, c) / c ** 0.5
ref = F.conv1d(x, w, padding=1); out = run(c, L, 3, w, x); print("mix center tap:", (out - ref).abs().max().item())
print("---- factors")
def run2(c, L, k, B, bias, nres, scale, cfg54=0):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, c, L, generator=g); w = torch.randn(c, c, k, generator=g) / (c * k) ** 0.5
    b = torch.randn(c, generator=g) if bias else None
    res = [torch.randn(B, c, L, generator=g) for _ in range(nres)]
    ref = F.conv1d(x.double(), w.double(), b.double() if bias else None, padding=(k - 1) // 2)
    for r in res: ref = ref + r.double()
    ref = ref * scale
    bm = lib.fh_wino54_tile_m(cfg54); cpad = -(-c // bm) * bm
    xd = x.to(DEV); out = torch.full_like(xd, float("nan")); rd = [r.to(DEV) for r in res]
    u = V.pack_wino54_weight(w, cpad).to(DEV)
    seg = V.make_wino_seg(xd, u, c, k); seg.ngrp = -(-k // 4)
    grp = V.make_wino_group([seg], b.to(DEV) if bias else None, rd, out, c, cpad, L, scale=scale)
    desc = hip.to_device_struct_array([grp], DEV)
    hip.check(lib.fh_conv_wino54_f32(desc.data_ptr(), 1, B, cpad, L, 1, 0, cfg54, st), "wino54")
    torch.cuda.synchronize()
    d = (out.cpu().double() - ref).abs()
    nanc = int(torch.isnan(out).sum())
    print(f"C={c} L={L} k={k} B={B} bias={bias} nres={nres} scale={scale}: err {d[~torch.isnan(d)].max().item() if nanc < d.numel() else float('nan'):.2e} nan {nanc}  bad cols {int((d.amax((0,1)) > 1e-4).sum())}/{L} bad rows {int((d.amax((0,2)) > 1e-4).sum())}/{c}")
run2(128, 320, 3, 1, False, 0, 1.0)
run2(128, 320, 3, 2, False, 0, 1.0)
run2(128, 640, 3, 1, False, 0, 1.0)
run2(128, 320, 3, 1, True, 0, 1.0)
run2(128, 320, 3, 1, False, 1, 1.0)
run2(128, 320, 3, 1, False, 0, 0.5)
run2(128, 320, 11, 1, False, 0, 1.0)
run2(128, 324, 3, 1, False, 0, 1.0)
run2(256, 320, 3, 1, False, 0, 1.0)
