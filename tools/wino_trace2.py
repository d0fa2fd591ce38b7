"""Fine-grained per-wave timeline of one Winograd conv launch: where a block's time goes outside its K loop.
Needs the trace2 variant (tools/exp/conv_wino_trace2.hip: stamps around descriptor fetch, prologue, K loop and the
four phases of every epilogue sub-tile):
    tools/build_variant.sh trace2 tools/exp/conv_wino_trace2.hip=conv_wino.hip
    FH_LIB_PATH=tools/abl/trace2.so python tools/wino_trace2.py <C> <L> <dil> [tile_cfg] [nres]"""
import sys, ctypes, torch, numpy as np
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V
c, L, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 1
DEV = torch.device('cuda:0'); KS = [11, 7, 3]
xs = [torch.randn(B, c, L, device=DEV) for _ in KS]
outs = [torch.empty(B, c, L, device=DEV) for _ in KS]
res = [torch.randn(B, c, L, device=DEV) for _ in KS]
ws = [torch.randn(c, c, k) * 0.02 for k in KS]
bs = [torch.randn(c, device=DEV) for _ in KS]
wcfg, wpad = V.pick_wino_tile(c)
if len(sys.argv) > 4:
    wcfg = int(sys.argv[4])
nres = int(sys.argv[5]) if len(sys.argv) > 5 else 1
ud = [V.pack_wino_weight(w, wpad).to(DEV) for w in ws]
gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [res[i]] * nres, outs[i], c, wpad, L) for i, k in enumerate(KS)]
dw = hip.to_device_struct_array(gw, DEV)
st = hip.stream()
lib = hip.lib()
lib.fh_debug_set_wino_trace.argtypes = [ctypes.c_void_p]
pm = 1 if d > 1 else 0
run = lambda: hip.check(lib.fh_conv_wino_f32(dw.data_ptr(), 3, B, wpad, L, d, pm, wcfg, st))
NW = 24
buf = torch.zeros(1 + NW * 200000, dtype=torch.int64, device=DEV)
# WARM launches straight before the traced one, nothing in between: the chip needs ~20 ms of load to reach its
# sustained clock (tools/wino_sustained.py); WARM=20 (12 ms, then a sync) is what the round-3 tables were taken with
import os
warm = int(os.environ.get("WARM", "20"))
# ALT=1: every warm-up conv launch is preceded by an activation launch of the same tensors (the model's alternation), and so is
# the traced one: on some boxes of the pool a conv launch that follows an activation launch is 14 % slower (tools/box_probe.sh)
if os.environ.get("ALT", "0") == "1":
    from flowhigh_amd import synth
    filt = synth.kaiser_sinc_filter().flatten().tolist()
    pa = dict(alpha=torch.rand(c, device=DEV) + 0.5, inv_beta=torch.rand(c, device=DEV) + 0.5, up=filt, down=filt)
    ya = [torch.empty(B, c, L, device=DEV) for _ in KS]
    ga = hip.to_device_struct_array([V.make_act_group(res[i], ya[i], pa) for i in range(3)], DEV)
    conv_only = run
    def run():
        hip.check(lib.fh_act1d_grouped_pm_f32(ga.data_ptr(), 3, B, c, L, 1, 1, st))
        conv_only()
for _ in range(warm): run()
if warm <= 20: torch.cuda.synchronize()
hip.check(lib.fh_debug_set_wino_trace(buf.data_ptr()))
run(); torch.cuda.synchronize()
hip.check(lib.fh_debug_set_wino_trace(0))
a = buf.cpu().numpy(); rec = a[1:].reshape(-1, NW); rec = rec[rec[:, 1] > 0]; n = rec.shape[0]
bid = rec[:, 0] & 0xffffffff
hw = (rec[:, 0] >> 32) & 0xffffff; xcc = (rec[:, 0] >> 56) & 0xf
cuid = xcc * 256 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 0xf)
t0 = rec[:, 1]; s0 = t0.min()
us = lambda col: (rec[:, col] - t0) / 100.0
sub_cols = [(7 + 3 * k if k < 3 else 19) for k in range(4)]
nsub = sum(1 for c_ in sub_cols if rec[0, c_] > 0)
print(f"C={c} L={L} d={d} cfg={wcfg} nres={nres}: waves {n}, blocks {len(set(bid.tolist()))}, launch {(rec[:, 2].max() - s0) / 100.0:.1f} us, sub-tiles {nsub}")
first = {}
for i in np.argsort(t0):
    first.setdefault(int(cuid[i]), int(bid[i]))
is_first = np.array([first[int(cuid[i])] == int(bid[i]) for i in range(n)])
loop = us(6) - us(18)
for name, m in (("first block of a CU", is_first), ("later blocks", ~is_first)):
    if not m.any():
        continue
    print(f"--- {name}: {m.sum() // 12} blocks")
    for lab, lo, hi in (("long K loop (k=11)", np.percentile(loop, 66), 1e9), ("short K loop (k=3)", 0, np.percentile(loop, 33))):
        mm = m & (loop >= lo) & (loop <= hi)
        if not mm.any():
            continue
        def med(x): return float(np.median(x[mm]))
        line = [f"start->desc {med(us(3)):.2f} | desc fetch {med(us(4) - us(3)):.2f} | first A arrive {med(us(16) - us(4)):.2f} | "
                f"slab load+LDS store {med(us(17) - us(16)):.2f} | barrier {med(us(18) - us(17)):.2f}", f"K loop {med(loop):.1f}"]
        prev = us(6)
        for k in range(nsub):
            c0_ = sub_cols[k]
            b1, b2, e = us(c0_), us(c0_ + 1), us(c0_ + 2)
            line.append(f"sub{k}: wait-barrier {med(b1 - prev):.2f} | loads+E write+barrier {med(b2 - b1):.2f} | read+A^T+store {med(e - b2):.2f}")
            prev = e
        line.append(f"tail {med(us(2) - prev):.2f}")
        line.append(f"TOTAL outside K loop {med(us(2) - loop):.2f} of {med(us(2)):.1f} us")
        print(f"  [{lab}, {mm.sum() // 12} blocks] " + "\n      ".join(line))
# spread of K-loop exit between the waves of a block
ends = {}
for i in range(n):
    ends.setdefault(int(bid[i]), []).append((rec[i, 6] - s0) / 100.0)
sp = np.array([max(v) - min(v) for v in ends.values()])
print(f"K-loop exit spread inside a block (latest - earliest wave): median {np.median(sp):.2f} us, p90 {np.percentile(sp, 90):.2f}")
clk = ((rec[:, 23] >> 8) / np.maximum(rec[:, 2] - rec[:, 1], 1)) * 100.0
print(f"shader clock: median {np.median(clk):.0f} MHz")
