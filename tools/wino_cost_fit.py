"""Block-time constants of the Winograd kernel for the launch model (vocoder._WINO_COST: block time = a * K + b us,
K = 16-channel x tap-group steps): launches of ~8 blocks per CU at two depths, fp32 and bf16 x 6 forms.
python tools/wino_cost_fit.py"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip, vocoder as V
DEV = torch.device('cuda:0')
st = hip.stream()
TILES = {0: (64, 512), 1: (96, 256), 4: (64, 256), 5: (32, 256), 6: (128, 256)}


def block_time(c, cfg, mode, k=11):
    bm, bt = TILES[cfg]
    wpad = c
    tiles = max(4, 2048 * bm // wpad)                    # blocks = (wpad / bm) * tiles ~ 2048
    L = tiles * bt
    x = torch.randn(1, c, L, device=DEV)
    out = torch.empty(1, c, L, device=DEV)
    b = torch.randn(c, device=DEV)
    u = V.pack_wino_weight(torch.randn(c, c, k) * 0.02, wpad)
    u = (V.split_bf3(u) if mode else u).to(DEV)
    g = [V.make_wino_group([V.make_wino_seg(x, u, c, k)], b, [], out, c, wpad, L)]
    dw = hip.to_device_struct_array(g, DEV)
    run = lambda: hip.check(hip.lib().fh_conv_wino_f32(dw.data_ptr(), 1, 1, wpad, L, 1, 0, cfg | mode, st))
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    blocks = (wpad // bm) * tiles
    return e0.elapsed_time(e1) * 200 * 256 / blocks      # us per block, 256 CUs, one block per CU at a time


for mode in (0, V.WINO_BF16X6):
    res = {}
    for cfg in TILES:
        t1, t2 = block_time(384, cfg, mode), block_time(768, cfg, mode)
        k1, k2 = 384 // 16 * 4, 768 // 16 * 4
        a = (t2 - t1) / (k2 - k1)
        res[cfg] = (round(a, 3), round(t1 - a * k1, 1))
    print("bf16x6" if mode else "fp32  ", res)
