"""Residual-stack launch shapes of the 10 s clip under every Winograd tile.  With the experiment build
(tools/exp/build_wino_variants.sh; FH_LIB_PATH=flowhigh_amd/lib/abl/winox.so) also tile 8 (conv_wino2.hip: 4-wave blocks,
six transform points per wave) and tile 9 (conv_wino3.hip: persistent workgroups of two 6-wave teams), each compared
bit for bit with tile 4.  python tools/wino2_bench.py [B] [frames] [C:up,...]   (profiles/r03_wino_block_shapes.txt)"""
import sys
import torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V

DEV = torch.device('cuda:0')
KS = [11, 7, 3]
st = hip.stream()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
TILES = {0: (64, 512), 1: (96, 256), 4: (64, 256), 6: (128, 256)}
if hip.lib().fh_wino_tile_m(8) > 0:          # experiment build
    TILES.update({8: (64, 256), 9: (64, 256)})


def time_launch(dw, ng, wpad, L, d, pm, cfg, reps=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    run = lambda: hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), ng, B, wpad, L, d, pm, cfg, st))
    for _ in range(3):
        run()
    ev[0].record()
    for _ in range(reps):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


def shape(c, L, d, nres, closing=False):
    pm = 1 if d > 1 else 0
    pitch = d * V.phase_len(L, d) if pm else L
    xs = [torch.randn(B, c, pitch, device=DEV) for _ in KS]
    outs = [torch.empty(B, c, pitch, device=DEV) for _ in KS]
    rs = [torch.randn(B, c, pitch, device=DEV) for _ in KS]
    bs = [torch.randn(c, device=DEV) for _ in KS]
    wpad = -(-c // 64) * 64
    if c % 64 and c % 96 == 0:
        wpad = c
    ud = [V.pack_wino_weight(torch.randn(c, c, k) * 0.02, wpad).to(DEV) for k in KS]
    if closing:
        gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k) for i, k in enumerate(KS)], bs[0], rs[:nres], outs[0],
                                c, wpad, L, scale=1.0 / 3)]
    else:
        gw = [V.make_wino_group([V.make_wino_seg(xs[i], ud[i], c, k)], bs[i], [rs[i]] * nres, outs[i], c, wpad, L)
              for i, k in enumerate(KS)]
    dw = hip.to_device_struct_array(gw, DEV, hip.WINO3_WS_BYTES)
    flops = sum(2.0 * c * c * 1.5 * -(-k // 3) * L * B for k in KS)
    row, ref = [], None
    for cfg, (bm, bt) in TILES.items():
        if wpad % bm:
            continue
        for o in outs:
            o.fill_(float("nan"))
        t = time_launch(dw, len(gw), wpad, L, d, pm, cfg)
        if cfg == 4:
            ref = [o.clone() for o in outs]
        same = ""
        if cfg in (8, 9) and ref is not None:
            eq = lambda a_, b_: torch.equal(torch.nan_to_num(a_, nan=7.0), torch.nan_to_num(b_, nan=7.0))
            same = " ==4" if all(eq(o, r) for o, r in zip(outs, ref)) else " DIFFERS from 4"
        row.append(f"cfg{cfg} {t:7.1f} us {flops / t / 1e6:6.1f} TF{same}")
    print(f"c={c:4d} L={L:6d} d={d} nres={nres} {'closing' if closing else 'stack  '} | " + " | ".join(row), flush=True)


SHAPES = [(int(a), int(b)) for a, b in (s.split(":") for s in sys.argv[3].split(","))] if len(sys.argv) > 3 else [(768, 5), (384, 20), (192, 60), (96, 120), (48, 240)]
for c, up in SHAPES:
    L = N * up
    shape(c, L, 1, 1)
    shape(c, L, 3, 0)
    shape(c, L, 5, 0)
    shape(c, L, 1, 3, closing=True)
