"""A few launches of one activation shape (for rocprofv3 --pmc).  python tools/act_one.py C L din dout"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, synth, vocoder as V
c, L, din, dout = (int(v) for v in sys.argv[1:5])
DEV = torch.device('cuda:0')
filt = synth.kaiser_sinc_filter().flatten().tolist()
n = max(L, max(d * V.phase_len(L, d) for d in (din, dout)))
xs = [torch.randn(1, c, n, device=DEV) for _ in range(3)]
ys = [torch.empty(1, c, n, device=DEV) for _ in range(3)]
p = dict(alpha=torch.rand(c, device=DEV) + 0.5, inv_beta=torch.rand(c, device=DEV) + 0.5, up=filt, down=filt)
g = hip.to_device_struct_array([V.make_act_group(xs[i], ys[i], p) for i in range(3)], DEV)
for _ in range(5):
    hip.check(hip.lib().fh_act1d_grouped_pm_f32(g.data_ptr(), 3, 1, c, L, din, dout, hip.stream()))
torch.cuda.synchronize()
