"""Block-time constants of the F(5,4) kernel for the launch model (vocoder._WINO_COST[WINO_F54 | t] = (a, b): block time =
a K + b us, K = 16-channel x 4-tap-group steps): launches of ~8 blocks per CU at two depths per tile height.
python tools/wino54_cost_fit.py [bf]      (bf: also the bf16 x 6 form of round 6)"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V
DEV = torch.device('cuda:0')
st = hip.stream()


def block_time(c, tile, k=11, bf=False):
    bm, bt = V._WINO_TILES[V.WINO_F54 | tile]
    wpad = c
    tiles = max(4, 2048 * bm // wpad)                    # blocks = (wpad / bm) * tiles ~ 2048
    L = tiles * bt
    x = torch.randn(1, c, L, device=DEV)
    out = torch.empty(1, c, L, device=DEV)
    b = torch.randn(c, device=DEV)
    u = V.pack_wino54_weight_any(torch.randn(c, c, k) * 0.02, wpad, bf).to(DEV)
    g = [V.make_wino_group([V.make_wino_seg(x, u, c, k, taps=4)], b, [], out, c, wpad, L)]
    dw = hip.to_device_struct_array(g, DEV)
    run = lambda: hip.check(hip.lib().fh_conv_wino54_f32(dw.data_ptr(), 1, 1, wpad, L, 1, 0, tile | (V.WINO_BF16X6 if bf else 0), st))
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    blocks = (wpad // bm) * tiles
    return e0.elapsed_time(e1) * 200 * 256 / blocks      # us per block, 256 CUs, one block per CU at a time


for bf in ([False, True] if "bf" in sys.argv[1:] else [False]):
    res = {}
    for tile in ((1, 2) if bf else (0, 1, 2)):
        t1, t2 = block_time(384, tile, bf=bf), block_time(768, tile, bf=bf)
        k1, k2 = 384 // 16 * 3, 768 // 16 * 3
        a = (t2 - t1) / (k2 - k1)
        bm = V._WINO_TILES[V.WINO_F54 | tile][0]
        # MT x 2 columns x (8 k-step MFMAs of 64 cycles | 6 piece-pair MFMAs of 32 cycles), 2 waves per SIMD, at 2.38 GHz
        floor = bm // 32 * 2 * (6 * 32 if bf else 8 * 64) * 2 / 2380.0
        res[tile] = (round(a, 3), round(t1 - a * k1, 1))
        print(f"{'bf16x6' if bf else 'fp32  '} tile {tile} ({bm} x 320): a = {a:.3f} us per K step (matrix-pipe floor {floor:.3f}: {floor / a:.2f}), "
              f"b = {t1 - a * k1:.1f} us")
    print(res)
