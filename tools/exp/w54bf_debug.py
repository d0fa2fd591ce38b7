"""Round 6 bring-up of the F(5,4) bf16 x 6 kernel: one case per (layout / loader) x tile height against float64."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V
DEV = torch.device('cuda:0')
for (c, k, d, L, B, pm) in [(192, 11, 1, 3932, 2, False), (192, 11, 1, 3933, 2, False), (384, 7, 5, 34, 1, False), (192, 7, 3, 999, 1, True),
                            (96, 7, 2, 468, 3, False), (384, 3, 1, 19, 2, False)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, c, L, generator=g)
    w = torch.randn(c, c, k, generator=g) / (c * k) ** 0.5
    b = torch.randn(c, generator=g)
    ref = F.conv1d(x.double(), w.double(), b.double(), dilation=d, padding=(k - 1) // 2 * d).float()
    xd = (V.to_phase_major(x, d) if pm else x).to(DEV)
    u = V.pack_wino54_weight(w, c)
    for cfg, uu in ((V.WINO_F54 | 1, u), (V.WINO_F54 | 1 | V.WINO_BF16X6, V.split_bf3(u)),
                    (V.WINO_F54 | 2 | V.WINO_BF16X6, V.split_bf3(u))):
        if c % V._WINO_TILES[cfg & (V.WINO_F54 | 15)][0]:
            continue
        ud, bd = uu.to(DEV), b.to(DEV)
        out = torch.full_like(xd, float("nan"))
        grp = V.make_wino_group([V.make_wino_seg(xd, ud, c, k, taps=4)], bd, [], out, c, c, L)
        keep = V.conv_wino([grp], B, c, L, d, DEV, cfg, phase_major=pm)
        torch.cuda.synchronize()
        got = V.from_phase_major(out.cpu(), d, L) if pm else out.cpu()
        err = (got - ref).abs()
        bad = (err > 1e-3).nonzero()
        print(f"c={c} k={k} d={d} L={L} B={B} pm={pm} cfg={cfg:#x}: max err {err.max().item():.3e}  bad {bad.shape[0]}"
              + (f" first {bad[0].tolist()} last {bad[-1].tolist()} rows {sorted(set(bad[:, 1].tolist()))[:8]}.." if bad.shape[0] else ""))
