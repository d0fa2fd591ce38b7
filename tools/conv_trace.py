"""Per-block timeline of the conv launches of one vocoder pass (debug hook fh_debug_set_conv_trace)."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from flowhigh_amd import hip, synth
from flowhigh_amd.vocoder import Vocoder
cfg = synth.SYNTH_CFG
sd = synth.make_vocoder_state_dict(cfg, 0)
voc = Vocoder(cfg, sd, 'cuda')
mel = (torch.randn(1, 1000, 256) * 2 - 3).cuda()
for _ in range(3):
    voc.forward(mel)
torch.cuda.synchronize()
buf = torch.zeros(1 + 4 * 200000, dtype=torch.int64, device='cuda')
hip.check(hip.lib().fh_debug_set_conv_trace(buf.data_ptr()))
voc.forward(mel)
torch.cuda.synchronize()
hip.check(hip.lib().fh_debug_set_conv_trace(0))
a = buf.cpu().numpy()
n = int(a[0]); rec = a[1:1 + 4 * n].reshape(n, 4)
t0, t1, steps = rec[:, 1], rec[:, 2], rec[:, 3]
order = np.argsort(t0); rec = rec[order]; t0, t1, steps = rec[:, 1], rec[:, 2], rec[:, 3]
hw = (rec[:, 0] >> 32) & 0xffffff; xcc = (rec[:, 0] >> 56) & 0xf
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7   # gfx9 HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
cuid = xcc * 256 + se * 16 + cu
# split launches by gaps in start time > 20 us (2000 ticks)
cuts = [0] + [i for i in range(1, n) if t0[i] - t0[i - 1] > 1500 and t0[i] > t1[:i].max() - 200] + [n]
print("launches found", len(cuts) - 1)
for a_, b_ in zip(cuts[:-1], cuts[1:]):
    s0, e1 = t0[a_:b_].min(), t1[a_:b_].max()
    dur = (e1 - s0) / 100.0
    nb = b_ - a_
    busy = (t1[a_:b_] - t0[a_:b_]).sum() / 100.0
    cus = len(set(cuid[a_:b_].tolist()))
    # time-weighted concurrency: sum of block durations / (dur * cus)
    ends = np.sort((t1[a_:b_] - s0) / 100.0)
    work = steps[a_:b_].sum()
    per_cu = np.bincount(np.unique(cuid[a_:b_], return_inverse=True)[1], weights=steps[a_:b_])
    print(f"blocks {nb:5d} dur {dur:8.1f} us  CUs {cus:3d}  avg conc/CU {busy/dur/cus:4.2f}  first end {ends[0]:7.1f} median end {ends[nb//2]:7.1f}  "
          f"steps/CU min {per_cu.min():.0f} mean {per_cu.mean():.0f} max {per_cu.max():.0f}  ideal_frac {per_cu.mean()/per_cu.max():.2f}")

def classes(li):
    a_, b_ = cuts[li], cuts[li + 1]
    s0 = t0[a_:b_].min()
    for ns in sorted(set(steps[a_:b_].tolist())):
        m = steps[a_:b_] == ns
        st = (t0[a_:b_][m] - s0) / 100.0; en = (t1[a_:b_][m] - s0) / 100.0
        print(f"   launch {li}: nsteps {ns:5d} blocks {m.sum():4d} start mean {st.mean():7.1f} (max {st.max():7.1f})  end mean {en.mean():7.1f} min {en.min():7.1f} max {en.max():7.1f}  dur mean {(en-st).mean():7.1f}")
for li in (2, 9, 16, 30):
    classes(li)

# which block ids share a CU (launch 2)
a_, b_ = cuts[2], cuts[3]
bid = (rec[a_:b_, 0] & 0xffffffff)
from collections import defaultdict
m = defaultdict(list)
for i in range(b_ - a_):
    m[int(cuid[a_ + i])].append(int(bid[i]))
for k in list(sorted(m))[:12]:
    print("   CU", k, sorted(m[k]))
