"""One fh_conv_wino_f32 configuration against float64 F.conv1d:
python tests/tools/wino_case.py c k d B L pm nres cfg [seed]"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from flowhigh_amd import vocoder as V
DEV = torch.device('cuda:0')
c, k, d, B, L, pm, nres, wcfg = (int(v) for v in sys.argv[1:9])
seed = int(sys.argv[9]) if len(sys.argv) > 9 else 0
g = torch.Generator().manual_seed(seed)
x = torch.randn(B, c, L, generator=g)
w = torch.randn(c, c, k, generator=g) / (c * k) ** 0.5
bias = torch.randn(c, generator=g)
res = [torch.randn(B, c, L, generator=g) for _ in range(nres)]
ref = (F.conv1d(x.double(), w.double(), bias.double(), dilation=d, padding=(k - 1) // 2 * d) + sum(r.double() for r in res)).float()
cpad = -(-c // 384) * 384 if wcfg in (1, 3) else -(-c // 128) * 128
conv = lambda t: (V.to_phase_major(t, d) if pm else t).to(DEV)
xd, rd = conv(x), [conv(r) for r in res]
out = torch.full_like(xd, float("nan"))
ud, bd = V.pack_wino_weight(w, cpad).to(DEV), bias.to(DEV)
grp = V.make_wino_group([V.make_wino_seg(xd, ud, c, k)], bd, rd, out, c, cpad, L)
keep = V.conv_wino([grp], B, cpad, L, d, DEV, wcfg, phase_major=bool(pm))
torch.cuda.synchronize()
got = V.from_phase_major(out.cpu(), d, L) if pm else out.cpu()
err = (got - ref).abs()
bad = (err > 1e-4).nonzero()
print(f"c={c} k={k} d={d} B={B} L={L} pm={pm} nres={nres} cfg={wcfg}: max err {err.max().item():.3e}, {len(bad)} bad "
      f"elements", (f"first {bad[0].tolist()} last {bad[-1].tolist()} rows {sorted(set(bad[:, 1].tolist()))[:8]} "
                    f"t range {bad[:, 2].min().item()}..{bad[:, 2].max().item()}" if len(bad) else ""))
