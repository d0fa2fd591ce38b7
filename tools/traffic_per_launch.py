"""HBM traffic of every conv launch of one bench step against its algorithmic bytes, from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 1` (tools/profile_round.sh); corrections as in
make_traffic_json.py.  usage: traffic_per_launch.py <fetch_dir> <write_dir>  (run anywhere: the plan is built on CPU)"""
import csv, glob, sys
sys.path.insert(0, '.')
from flowhigh_amd import synth
from flowhigh_amd.vocoder import Vocoder

fetch_dir, write_dir = sys.argv[1], sys.argv[2]
names = ("conv_mfma_kernel", "conv_wino_kernel", "conv_wino54_kernel")


def series(d, name):
    f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0]
    rows = [(int(r['Dispatch_Id']), float(r['Counter_Value'])) for r in csv.DictReader(open(f))
            if r['Counter_Name'] == name and any(k in r['Kernel_Name'] for k in names)]
    rows.sort()
    return [v for _, v in rows]


fs, ws = series(fetch_dir, 'FETCH_SIZE'), series(write_dir, 'WRITE_SIZE')
cfg = synth.SYNTH_CFG
voc = Vocoder(cfg, synth.make_vocoder_state_dict(cfg, 0), 'cpu')
p = voc.plan(1, 1000)
convs = [(s, m) for s, m in zip(p['steps'], p['meta']) if s[0] in ('conv', 'wino')]
n = len(convs)
assert len(fs) % n == 0 and len(fs) == len(ws), (len(fs), len(ws), n)
fs, ws = fs[-n:], ws[-n:]                   # the last (timed) step
print(f"{'#':>3} {'kind':>5} {'C':>5} {'L':>7} {'read MB':>8} {'alg':>7} {'x':>5} {'write MB':>9} {'alg':>7} {'weights MB':>10}")
tot = [0.0] * 4
for i, ((s, (key, groups)), f, w) in enumerate(zip(convs, fs, ws)):
    rd, wr = 2.0 * f * 1024 / 1e6, w * 1024 / 1e6
    kind = s[0]
    if s[0] == 'wino':
        _, _d, ng, wpad, length, dil, _fl, wcfg, pm, bb = s
        kind = 'winoR' if wcfg & 32 else 'wino'
        a_in = sum(g.seg[j].cin * length * 4 for g in groups for j in range(g.nseg)) * (1 if bb == 1 and ng > 3 else 1)
        a_res = sum(g.nres * g.cout * length * 4 for g in groups)
        a_out = sum(g.cout * length * 4 * max(1, g.out_stride) / max(1, g.out_stride) for g in groups)
        wts = sum(g.seg[j].cin * g.seg[j].ngrp * 6 * wpad * 4 for g in groups for j in range(g.nseg))
        c = groups[0].seg[0].cin
    else:
        _, _d, ng, cpad, n_len, tcfg, ck, _fl = s
        length = n_len
        a_in = sum(g.seg[j].cin * g.lin * 4 for g in groups for j in range(g.nseg))
        a_res = sum(g.nres * g.cout * n_len * 4 for g in groups)
        a_out = sum(g.cout * n_len * 4 for g in groups)
        wts = sum(g.seg[j].cin * g.seg[j].ntaps * cpad * 4 for g in groups for j in range(g.nseg))
        c = groups[0].seg[0].cin
    # transposed-conv phase groups read the same input: count it once
    if key[1] == -1:
        a_in /= len(groups)
    alg_r, alg_w = (a_in + a_res + wts) / 1e6, a_out / 1e6
    tot[0] += rd; tot[1] += alg_r; tot[2] += wr; tot[3] += alg_w
    print(f"{i:3d} {kind:>5} {c:5d} {length:7d} {rd:8.1f} {alg_r:7.1f} {rd / alg_r:5.2f} {wr:9.1f} {alg_w:7.1f} {wts / 1e6:10.1f}")
print(f"total read {tot[0]:.0f} MB (algorithmic incl. weights once {tot[1]:.0f}: x {tot[0] / tot[1]:.2f}), written {tot[2]:.0f} MB "
      f"(algorithmic {tot[3]:.0f}: x {tot[2] / tot[3]:.2f}); per launch {(tot[0] + tot[2]) / n:.1f} MB against {(tot[1] + tot[3]) / n:.1f} MB = x {(tot[0] + tot[2]) / (tot[1] + tot[3]):.2f}")
