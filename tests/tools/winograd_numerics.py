"""CPU experiment: end-to-end error of the vocoder when its k = 3/7/11 convs run as Winograd
F(4,3) (taps in groups of 3, sums over channels and groups in the transform domain, fp32)."""
import sys, math, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from flowhigh_amd import synth
from oracle import ref_cpu

Bt = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                   [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
Gm = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                   [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)

def wino_conv1d(x, w, b, d):
    B, Ci, L = x.shape
    Co, _, k = w.shape
    c = (k - 1) // 2
    G = -(-k // 3)
    wpad = F.pad(w.double(), (0, 3 * G - k))
    dt = x.dtype
    U = [torch.einsum('xj,ocj->xoc', Gm, wpad[:, :, 3 * g:3 * g + 3]).to(dt) for g in range(G)]
    y = torch.empty(B, Co, L, dtype=dt)
    for p in range(d):
        xp = x[..., p::d]
        Lp = xp.shape[-1]
        T = -(-Lp // 4)
        xq = F.pad(xp, (c, 4 * T + 3 * G + 8 - Lp))
        M = None
        for g in range(G):
            tiles = xq[..., 3 * g:].unfold(-1, 6, 4)[:, :, :T, :]            # [B,Ci,T,6]
            V = torch.einsum('xj,bctj->xbct', Bt.to(dt), tiles)
            Mg = torch.matmul(U[g].unsqueeze(1), V)                           # [6,B,Co,T]
            M = Mg if M is None else M + Mg
        Y = torch.einsum('ix,xbot->boti', At.to(dt), M).reshape(B, Co, 4 * T)[..., :Lp]
        y[..., p::d] = Y
    return y + b.view(1, -1, 1)

x = torch.randn(1, 96, 3000); w = torch.randn(64, 96, 11) / math.sqrt(96 * 11); b = torch.randn(64)
for d in (1, 3, 5):
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=5 * d, dilation=d)
    e_w = (wino_conv1d(x, w, b, d).double() - ref).abs().max().item()
    e_d = (F.conv1d(x, w, b, padding=5 * d, dilation=d).double() - ref).abs().max().item()
    print(f"k=11 d={d}: winograd fp32 err {e_w:.2e}, direct fp32 err {e_d:.2e}, |y|max {ref.abs().max():.2f}")

# end to end through the oracle's vocoder
orig = F.conv1d
def patched(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    k = w.shape[-1]
    if groups == 1 and stride == 1 and k in (3, 7, 11) and w.shape[0] == w.shape[1] and x.dtype == torch.float32 \
            and padding == (k - 1) // 2 * dilation:
        return wino_conv1d(x, w, b, dilation)
    return orig(x, w, b, stride, padding, dilation, groups)

for name, cfg, n in (("TINY", synth.TINY_CFG, 50), ("SYNTH", synth.SYNTH_CFG, 30)):
    sd = synth.make_state_dict(cfg, 0)
    mel = torch.randn(1, 256, n) * 2 - 3
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref = ref_cpu.bigvgan_forward(sd64, cfg, mel.double())
    d32 = ref_cpu.bigvgan_forward(sd, cfg, mel)
    F.conv1d = patched
    try:
        w32 = ref_cpu.bigvgan_forward(sd, cfg, mel)
    finally:
        F.conv1d = orig
    print(f"{name}: direct fp32 vs fp64 {(d32.double() - ref).abs().max():.2e}; winograd fp32 vs fp64 {(w32.double() - ref).abs().max():.2e}; |wav|max {ref.abs().max():.3f}")
