import sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
from flowhigh_amd import vocoder as V
DEV='cuda:0'
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
for (C,k,d,L,B) in [(48,11,5,2000,1),(48,3,1,644,2),(40,9,4,777,1),(48,11,1,320,1),(32,5,2,1280,1)]:
    x, w, b = rnd(B, C, L, seed=1), rnd(C, C, k, seed=2, scale=1.0 / (C * k) ** 0.5), rnd(C, seed=3)
    r1 = rnd(B, C, L, seed=4)
    ref = ((F.conv1d(x.double(), w.double(), None, dilation=d, padding=(k-1)//2*d) + b.double()[None,:,None] + r1.double()) * 0.5).float()
    xd, rd, bd = x.to(DEV), r1.to(DEV), b.to(DEV)
    out = torch.full_like(xd, float("nan"))
    ud = V.pack_amp_weight(w, C).to(DEV)
    g = V.make_amp_group([V.make_amp_seg(xd, ud, k)], bd, [rd], out, L, scale=0.5)
    keep = V.amp_actconv([g], B, C, d, DEV)
    torch.cuda.synchronize()
    err = (out.cpu()-ref).abs()
    bad = (err > 2e-5) | torch.isnan(err)
    print((C,k,d,L,B), 'max err', float(err[~torch.isnan(err)].max()), 'nan', int(torch.isnan(err).sum()), 'bad', int(bad.sum()))
    if bad.any():
        idx = bad.nonzero()
        print('  bad channels', sorted(set(idx[:,1].tolist()))[:60])
        ts = idx[:,2]
        print('  bad t range', int(ts.min()), int(ts.max()), 'count by tile(300)', torch.bincount(ts//300)[:10].tolist())
