"""CPU experiment: error of the vocoder's k = 3 / 7 / 11 convs as 1-D minimal filtering F(m, r) in fp32 -- the form the
product uses, F(4,3), and the larger tiles that would execute fewer multiply-adds (F(6,3): 8 points, F(4,4): 7,
F(5,4): 8, F(6,4): 9) -- per conv and end to end through the oracle's vocoder, both against float64.

    python tests/tools/winograd_numerics.py            # table for all forms / point sets (profiles/r04_winograd_numerics.txt)

Transforms by Toom-Cook over the points a_0 .. a_(n-2) and infinity, n = m + r - 1, in exact rationals:
    y = A^T [ (G g) .* (B^T d) ],   A^T = V_m^T,  G = D^-1 V_r,  B^T = D (V_n^-1)^T,   V_k = [a_i^j] (row [0 .. 0 1] for inf),
D = diag(prod_(k != i) (a_i - a_k)) (the wincnn scaling: B^T gets small integers / simple fractions).  Taps are walked
in groups of r, products summed over channels and groups in the transform domain, fp32 everywhere except the weight
transform (float64 on the host, as the product does)."""
import math
import sys
from fractions import Fraction as Fr

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from flowhigh_amd import synth          # noqa: E402
from oracle import ref_cpu              # noqa: E402


def _inv(m):
    n = len(m)
    a = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(m)]
    for c in range(n):
        p = next(r for r in range(c, n) if a[r][c] != 0)
        a[c], a[p] = a[p], a[c]
        a[c] = [v / a[c][c] for v in a[c]]
        for r in range(n):
            if r != c and a[r][c] != 0:
                a[r] = [v - a[r][c] * w for v, w in zip(a[r], a[c])]
    return [row[n:] for row in a]


def toom_cook(m, r, pts):
    """(A^T [m x n], G [n x r], B^T [n x n]) as float64 tensors; pts = the n - 1 finite points."""
    n = m + r - 1
    pts = [Fr(p) for p in pts]
    assert len(pts) == n - 1 and len(set(pts)) == n - 1

    def vand(k):
        return [[p ** j for j in range(k)] for p in pts] + [[Fr(int(j == k - 1)) for j in range(k)]]
    f = [math.prod((pts[i] - pts[k]) for k in range(n - 1) if k != i) for i in range(n - 1)] + [Fr(1)]
    vinv = _inv(vand(n))                                   # s = V_n^-1 values
    bt = [[f[i] * vinv[j][i] for j in range(n)] for i in range(n)]
    g = [[v / f[i] for v in row] for i, row in enumerate(vand(r))]
    at = [[vand(m)[i][j] for i in range(n)] for j in range(m)]
    t = lambda mat: torch.tensor([[float(v) for v in row] for row in mat], dtype=torch.float64)
    return t(at), t(g), t(bt)


def check_identity(m, r, pts):
    at, g, bt = toom_cook(m, r, pts)
    d, w = torch.randn(m + r - 1, dtype=torch.float64), torch.randn(r, dtype=torch.float64)
    y = at @ ((g @ w) * (bt @ d))
    ref = torch.stack([sum(w[j] * d[i + j] for j in range(r)) for i in range(m)])
    assert (y - ref).abs().max() < 1e-9, (m, r, pts)


def wino_conv1d(x, w, b, d, form):
    m, r, pts = form
    at, gm, bt = toom_cook(m, r, pts)
    n = m + r - 1
    B, Ci, L = x.shape
    Co, _, k = w.shape
    c = (k - 1) // 2
    G = -(-k // r)
    wpad = F.pad(w.double(), (0, r * G - k))
    dt = x.dtype
    U = [torch.einsum('xj,ocj->xoc', gm, wpad[:, :, r * g:r * g + r]).to(dt) for g in range(G)]
    y = torch.empty(B, Co, L, dtype=dt)
    for p in range(d):
        xp = x[..., p::d]
        Lp = xp.shape[-1]
        T = -(-Lp // m)
        xq = F.pad(xp, (c, m * T + r * G + n - Lp))
        M = None
        for g in range(G):
            tiles = xq[..., r * g:].unfold(-1, n, m)[:, :, :T, :]            # [B,Ci,T,n]
            V = torch.einsum('xj,bctj->xbct', bt.to(dt), tiles)
            Mg = torch.matmul(U[g].unsqueeze(1), V)                           # [n,B,Co,T]
            M = Mg if M is None else M + Mg
        Y = torch.einsum('ix,xbot->boti', at.to(dt), M).reshape(B, Co, m * T)[..., :Lp]
        y[..., p::d] = Y
    return y + b.view(1, -1, 1)


H = Fr(1, 2)
FORMS = {
    "F(4,3) 0 +-1 +-2 inf (product)": (4, 3, [0, 1, -1, 2, -2]),
    "F(4,3) 0 +-1 +-1/2 inf": (4, 3, [0, 1, -1, H, -H]),
    "F(6,3) 0 +-1 +-2 +-1/2 inf": (6, 3, [0, 1, -1, 2, -2, H, -H]),
    "F(6,3) 0 +-1 +-1/2 +-3/2 inf": (6, 3, [0, 1, -1, H, -H, Fr(3, 2), -Fr(3, 2)]),
    "F(4,4) 0 +-1 +-1/2 2 inf": (4, 4, [0, 1, -1, H, -H, 2]),
    "F(4,4) 0 +-1 +-2 -1/2 inf": (4, 4, [0, 1, -1, 2, -2, -H]),
    "F(5,4) 0 +-1 +-2 +-1/2 inf": (5, 4, [0, 1, -1, 2, -2, H, -H]),
    "F(6,4) 0 +-1 +-2 +-1/2 -1/4 inf": (6, 4, [0, 1, -1, 2, -2, H, -H, -Fr(1, 4)]),
    "F(6,4) 0 +-1 +-2 +-1/2 3/2 inf": (6, 4, [0, 1, -1, 2, -2, H, -H, Fr(3, 2)]),
    "F(6,4) +-1 +-2 +-1/2 +-3/2 inf": (6, 4, [1, -1, 2, -2, H, -H, Fr(3, 2), -Fr(3, 2)]),
    "F(6,4) +-1/2 +-1 +-3/2 +-3/4 inf": (6, 4, [1, -1, H, -H, Fr(3, 2), -Fr(3, 2), Fr(3, 4), -Fr(3, 4)]),
}


def macs(form, ks=(3, 7, 11)):
    m, r, _ = form
    return [(m + r - 1) / m * -(-k // r) for k in ks]


def main():
    torch.manual_seed(0)
    for f in FORMS.values():
        check_identity(*f)
    x = torch.randn(1, 96, 3000)
    ws = {k: torch.randn(64, 96, k) / math.sqrt(96 * k) for k in (3, 7, 11)}
    b = torch.randn(64)
    print("per conv (96 -> 64 channels, N(0,1) input, unit-variance output): max |err| vs float64; direct fp32 is ~4e-7")
    for name, form in FORMS.items():
        errs = []
        for k in (3, 7, 11):
            for d in (1, 5):
                ref = F.conv1d(x.double(), ws[k].double(), b.double(), padding=(k - 1) // 2 * d, dilation=d)
                errs.append((wino_conv1d(x, ws[k], b, d, form).double() - ref).abs().max().item())
        mm = macs(form)
        print(f"  {name:36s} MACs/output k=3/7/11: {mm[0]:.2f} {mm[1]:.2f} {mm[2]:.2f} (sum {sum(mm):5.2f})   "
              f"err k=3: {max(errs[0:2]):.1e}  k=7: {max(errs[2:4]):.1e}  k=11: {max(errs[4:6]):.1e}")

    orig = F.conv1d
    cases = (("TINY", synth.TINY_CFG, 50), ("SYNTH", synth.SYNTH_CFG, 30))
    only = sys.argv[1:]
    print("end to end (oracle vocoder, every k = 3 / 7 / 11 residual-stack conv in the given form, fp32) vs float64:")
    for cname, cfg, nfr in cases:
        sd = synth.make_state_dict(cfg, 0)
        mel = torch.randn(1, 256, nfr, generator=torch.Generator().manual_seed(1)) * 2 - 3
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        ref = ref_cpu.bigvgan_forward(sd64, cfg, mel.double())
        d32 = ref_cpu.bigvgan_forward(sd, cfg, mel)
        print(f"  {cname}: |wav|max {ref.abs().max():.3f}; direct fp32 {(d32.double() - ref).abs().max():.2e}")
        for name, form in FORMS.items():
            if only and not any(o in name for o in only):
                continue

            def patched(x_, w_, b_=None, stride=1, padding=0, dilation=1, groups=1, form=form):
                k = w_.shape[-1]
                if groups == 1 and stride == 1 and k in (3, 7, 11) and w_.shape[0] == w_.shape[1] \
                        and x_.dtype == torch.float32 and padding == (k - 1) // 2 * dilation:
                    return wino_conv1d(x_, w_, b_, dilation, form)
                return orig(x_, w_, b_, stride, padding, dilation, groups)
            F.conv1d = patched
            try:
                w32 = ref_cpu.bigvgan_forward(sd, cfg, mel)
            finally:
                F.conv1d = orig
            print(f"    {name:36s} {(w32.double() - ref).abs().max():.2e}")


if __name__ == "__main__":
    main()
