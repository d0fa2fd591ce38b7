"""Phase timeline of the F(5,4) kernel's blocks (tools/exp/make_w54_trace.py build):
FH_LIB_PATH=tools/abl/w54trace.so python tools/exp/w54_trace.py [cout] [cin] [len] [nres]"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V

DEV = torch.device('cuda:0')
cout = int(sys.argv[1]) if len(sys.argv) > 1 else 96
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 96
L = int(sys.argv[3]) if len(sys.argv) > 3 else 120000
nres = int(sys.argv[4]) if len(sys.argv) > 4 else 0
KS = [11, 7, 3]
st = hip.stream()
lib = hip.lib()
cfg54 = 0 if cout % 128 == 0 else 1 if cout % 96 == 0 else 3 if cout % 48 == 0 else 2
bm = lib.fh_wino54_tile_m(cfg54)
cpad = -(-cout // bm) * bm
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
nblk = 1 << 14
tr = torch.zeros(nblk * 16, dtype=torch.int64, device=DEV)
raw = lib
raw.fh_w54_set_trace.argtypes = [ctypes.c_void_p]; raw.fh_w54_set_trace.restype = ctypes.c_int
for rep in range(3):
    tr.zero_()
    assert raw.fh_w54_set_trace(tr.data_ptr()) == 0
    hip.check(lib.fh_conv_wino54_f32(d.data_ptr(), 3, 1, cpad, L, 1, 0, cfg54, st), "w54")
    torch.cuda.synchronize()
raw_t = tr.cpu().view(nblk, 16)
live = raw_t[:, 10] > 0
hw = raw_t[live][:, 11]
t = raw_t[live].double() * 0.01          # us (100 MHz)
t0 = t[:, 0].min()
print(f"cout {cout} cin {cin} len {L} nres {nres}: {int(live.sum())} blocks; launch span {float(t[:, 10].max() - t0):.1f} us")
names = ["map->setup", "setup->loads issued", "loads->slab stored+barrier", "K loop", "epilogue requests", "round 0: E write + A^T + Y write",
         "round 0: barrier", "round 0: stores issued", "rounds 1..", "drain vmcnt"]
for ngrp_name, sel in (("all", slice(None)),):
    for i, nm in enumerate(names):
        dt = t[:, i + 1] - t[:, i]
        print(f"  {nm:36s} mean {float(dt.mean()):7.2f} us   median {float(dt.median()):7.2f}   p90 {float(dt.quantile(0.9)):7.2f}")
    tot = t[:, 10] - t[:, 0]
    print(f"  {'block total':36s} mean {float(tot.mean()):7.2f} us   median {float(tot.median()):7.2f}")
# gaps between consecutive blocks on a CU cannot be seen from here; the sum of block times against span * CUs:
print(f"  sum of block times / (256 CUs x span) = {float((t[:, 10] - t[:, 0]).sum() / (256 * (t[:, 10].max() - t0))):.3f}")

# idle time of a CU between two of its blocks: blocks grouped by (XCC, SE, CU) of HW_ID (gfx9 layout: CU_ID bits 11:8, SH 12, SE 15:13)
cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xff)
gaps = []
for c in cu.unique():
    tt = t[cu == c]
    tt = tt[tt[:, 0].argsort()]
    if len(tt) > 1:
        gaps.append(tt[1:, 0] - tt[:-1, 10])
g = torch.cat(gaps)
print(f"  {len(cu.unique())} CUs seen; gap between a block's last stamp and the next block's first on the same CU: mean {float(g.mean()):.2f} us  median {float(g.median()):.2f}  p90 {float(g.quantile(0.9)):.2f}")
first = torch.stack([t[cu == c][:, 0].min() for c in cu.unique()])
last = torch.stack([t[cu == c][:, 10].max() for c in cu.unique()])
print(f"  first block start after launch start: mean {float((first - t0).mean()):.2f} us; CU done before the launch's end: mean {float((t[:, 10].max() - last).mean()):.2f} us  max {float((t[:, 10].max() - last).max()):.2f}")
xcc = (hw >> 32) & 0xf
end = t[:, 10].max()
print("  per XCC: blocks, busy fraction of its 32 CUs over the span, last block end before the launch's end (us):")
for x in xcc.unique():
    m = xcc == x
    busy = float((t[m][:, 10] - t[m][:, 0]).sum() / (len(cu[m].unique()) * (end - t0)))
    print(f"    xcc {int(x)}: {int(m.sum()):4d} blocks  busy {busy:.3f}  ends {float(end - t[m][:, 10].max()):6.1f} us early; its CUs idle at the end: mean {float((end - torch.stack([t[cu == c][:, 10].max() for c in cu[m].unique()])).mean()):.1f}")
