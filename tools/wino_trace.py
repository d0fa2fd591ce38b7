"""Per-wave timeline of one Winograd conv launch (debug hook fh_debug_set_wino_trace)."""
import sys, ctypes, torch, numpy as np
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V
c, L, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 1
DEV = torch.device('cuda:0'); KS = [11, 7, 3]
xs = [torch.randn(B, c, L, device=DEV) for _ in KS]
outs = [torch.empty(B, c, L, device=DEV) for _ in KS]
ws = [torch.randn(c, c, k) * 0.02 for k in KS]
bs = [torch.randn(c, device=DEV) for _ in KS]
wcfg, wpad = V.pick_wino_tile(c)
if len(sys.argv) > 4:
    wcfg = int(sys.argv[4])
ud = [V.pack_wino_weight(w, wpad).to(DEV) for w in ws]
gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [], outs[i], c, wpad, L) for i, k in enumerate(KS)]
dw = hip.to_device_struct_array(gw, DEV)
st = hip.stream()
lib = hip.lib()
lib.fh_debug_set_wino_trace.argtypes = [ctypes.c_void_p]
run = lambda: hip.check(lib.fh_conv_wino_f32(dw.data_ptr(), 3, B, wpad, L, d, 0, wcfg, st))
for _ in range(3): run()
torch.cuda.synchronize()
buf = torch.zeros(1 + 5 * 100000, dtype=torch.int64, device=DEV)
hip.check(lib.fh_debug_set_wino_trace(buf.data_ptr()))
run(); torch.cuda.synchronize()
hip.check(lib.fh_debug_set_wino_trace(0))
a = buf.cpu().numpy(); n = int(a[0]); rec = a[1:1 + 5 * n].reshape(n, 5)
t0, t1, xi = rec[:, 1], rec[:, 2], rec[:, 3] & 0xff
pro = ((rec[:, 3] >> 8) & 0xfffffff) / 100.0; epi = (t1 - t0) / 100.0 - ((rec[:, 3] >> 36) & 0xfffffff) / 100.0
hw = (rec[:, 0] >> 32) & 0xffffff; xcc = (rec[:, 0] >> 56) & 0xf
bid = rec[:, 0] & 0xffffffff
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7
cuid = xcc * 256 + se * 16 + cu
s0 = t0.min(); dur = (t1.max() - s0) / 100.0
print(f"waves {n} blocks {len(set(bid.tolist()))} CUs {len(set(cuid.tolist()))} launch {dur:.1f} us")
busy = (t1 - t0).sum() / 100.0
print(f"avg resident waves per CU {busy / dur / len(set(cuid.tolist())):.2f}")
# per-SIMD wave time
per = {}
for i in range(n):
    per[(int(cuid[i]), int(simd[i]))] = per.get((int(cuid[i]), int(simd[i])), 0) + (t1[i] - t0[i]) / 100.0
v = np.array(list(per.values()))
print(f"wave-time per SIMD: min {v.min():.0f} mean {v.mean():.0f} max {v.max():.0f} us  (launch {dur:.0f})")
# block durations by start order
blocks = {}
for i in range(n):
    b_ = int(bid[i]); e = blocks.setdefault(b_, [1e18, 0, int(cuid[i]), []])
    e[0] = min(e[0], (t0[i] - s0) / 100.0); e[1] = max(e[1], (t1[i] - s0) / 100.0); e[3].append(int(simd[i]))
bl = sorted(blocks.items(), key=lambda kv: kv[1][0])
durs = np.array([e[1] - e[0] for _, e in bl]); starts = np.array([e[0] for _, e in bl])
for lo, hi in ((0, 5), (5, 50), (50, 200), (200, 400), (400, 600), (600, 800), (800, 1000), (1000, 1300)):
    pass
for lo, hi in ((0, 5), (5, 50), (50, 200), (200, 400), (400, 600), (600, 800), (800, 1000), (1000, 1300)):
    m = (starts >= lo) & (starts < hi)
    if m.sum(): print(f"  blocks started in [{lo},{hi}) us: {m.sum():4d}  dur mean {durs[m].mean():7.1f} min {durs[m].min():7.1f} max {durs[m].max():7.1f}")
from collections import Counter
print("SIMD placement patterns of blocks:", Counter(tuple(sorted(Counter(e[3]).values())) for _, e in bl).most_common(5))
percu = Counter(e[2] for _, e in bl)
print("blocks per CU: min", min(percu.values()), "max", max(percu.values()))
# concurrency on one CU
cu0 = bl[0][1][2]
ev = sorted((e[0], e[1], b_) for b_, e in bl if e[2] == cu0)
print("CU", cu0, [(round(a_), round(b_), c_) for a_, b_, c_ in ev])
# start / end of the blocks by launch-order decile (heavy groups come first in launch order)
order = sorted(blocks.items())
nb = len(order)
for q in range(10):
    part = order[q * nb // 10:(q + 1) * nb // 10]
    if part:
        print(f"  bids {part[0][0]:5d}..{part[-1][0]:5d}: start {np.mean([e[0] for _, e in part]):7.1f} "
              f"(max {max(e[0] for _, e in part):7.1f})  dur {np.mean([e[1] - e[0] for _, e in part]):7.1f}  "
              f"end max {max(e[1] for _, e in part):7.1f}")
idle = Counter()
for cu_, n_ in percu.items():
    idle[n_] += 1
print("CUs by number of blocks run:", dict(idle))
heavy = bid < nb // 4
print(f"first quarter of the blocks: prologue {pro[heavy].mean():.1f} us (max {pro[heavy].max():.1f}), epilogue "
      f"{epi[heavy].mean():.1f} us (max {epi[heavy].max():.1f}); all blocks: prologue {pro.mean():.1f}, epilogue {epi.mean():.1f}")
loop = (t1 - t0) / 100.0 - pro - epi
print(f"main loop of the first quarter of the blocks: {loop[heavy].mean():.1f} us (min {loop[heavy].min():.1f}, max {loop[heavy].max():.1f})")
mhz = rec[:, 4] / np.maximum(t1 - t0, 1) * 100.0
print(f"shader clock over the blocks (s_memtime / s_memrealtime): mean {mhz.mean():.0f} MHz, min {mhz.min():.0f}, max {mhz.max():.0f}")
