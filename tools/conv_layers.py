"""Per-launch table of the vocoder's conv launches: tile, blocks, algorithmic GFLOP, us, TFLOP/s.
Run on the GPU box: python tools/conv_layers.py [batch] [frames] [bypanel]"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import synth
from flowhigh_amd import vocoder as V
from flowhigh_amd.vocoder import Vocoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
TILES = {0: (128, 128), 1: (192, 128), 2: (96, 256), 3: (64, 256), 4: (32, 512), 5: (128, 64), 6: (96, 128)}
cfg = synth.SYNTH_CFG
if len(sys.argv) > 3 and sys.argv[3] == 'bypanel':         # A/B: every Winograd launch with the by-weight-panel block mapping
    V.wino_block_mapping = lambda *a: 0
import os
if os.environ.get("TILES_OFF"):                                # A/B: plan tile ids the launch model may not pick (e.g. 257 = F(5,4) 96 x 320)
    V._WINO_TILES_OFF.update(int(t) for t in os.environ["TILES_OFF"].split(","))
voc = Vocoder(cfg, synth.make_state_dict(cfg, 0), 'cuda:0')
p = voc.plan(B, N)
mel = torch.randn(B, N, cfg["num_mels"], generator=torch.Generator().manual_seed(0)) * 2.0 - 3.0
p["mel_in"].copy_(mel.transpose(1, 2).to('cuda:0'))
for _ in range(3):
    voc.run(p)
torch.cuda.synchronize()
convs = [s for s in p['steps'] if s[0] in ('conv', 'wino', 'convt', 'amp')]
launches = p["conv_launches"]                     # (family, executed FLOPs, algorithmic FLOPs) per conv launch: planner
assert len(convs) == len(launches)
acc = [0.0] * len(convs)
R = 10
for _ in range(R):
    voc.conv_timing = []
    voc.run(p)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(voc.conv_timing):
        acc[i] += a.elapsed_time(b) * 1e3 / R
voc.conv_timing = None
tot_f = tot_e = tot_t = 0.0
print(f"{'#':>3} {'family':>14} {'tile':>8} {'grp':>3} {'cpad':>5} {'n_len':>7} {'d':>2} {'blocks':>6} {'alg GFLOP':>9} {'exec GFLOP':>10} {'us':>8} {'exec TF/s':>9}")
for i, (s, (fam, ex, fl)) in enumerate(zip(convs, launches)):
    d_ = 1
    if s[0] == 'wino':
        _, d, ng, cpad, n_len, d_, _fl, wcfg, _pm = s[:9]
        f54, wcfg = wcfg & V.WINO_F54, wcfg & 15
        bm, bn = V._WINO_TILES[wcfg | f54]
        blocks = ng * s[9] * (cpad // bm) * V.wino_n_tiles(wcfg | f54, n_len, d_, _pm)
    elif s[0] == 'amp':
        _, d, ng, tiles, nt, c, d_ = s[:7]
        bn = V.hip.lib().fh_narrow_tile_len() if s[9] & V.AMP_DIRECT else V.amp_tile_len(d_)
        bm, cpad, n_len, blocks = c, c, 0, nt
    else:
        _, d, ng, cpad, n_len, tcfg = s[:6]
        bm, bn = TILES[tcfg]
        blocks = ng * B * (cpad // bm) * -(-n_len // bn)
    tot_f += fl; tot_e += ex; tot_t += acc[i]
    print(f"{i:3d} {fam:>14} {bm:>4}x{bn:<3} {ng:3d} {cpad:5d} {n_len:7d} {d_:2d} {blocks:6d} {fl/1e9:9.2f} {ex/1e9:10.2f} {acc[i]:8.1f} {ex/acc[i]/1e6:9.1f}")
print(f"total {tot_f/1e9:.1f} GFLOP algorithmic, {tot_e/1e9:.1f} executed, {tot_t/1e3:.3f} ms: {tot_e/tot_t/1e6:.1f} TF/s executed (fp32-equivalent)")
