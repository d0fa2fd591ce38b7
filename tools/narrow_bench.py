"""Time the two narrow-stage conv kernels on stage-sized launches: fh_amp_actconv_f32 (fp32-MFMA Winograd F(5,4), amp_fused.hip)
and fh_narrow_conv_bf16x6_f32 (direct bf16 x 6, narrow_bf.hip).
    python tools/narrow_bench.py [C=24] [L=480000] [reps=40]
Launches of three groups with bias and one residual: the model's (k = 11 / 7 / 3) at d = 1 / 3 / 5, then three groups of the SAME
k (k = 1, 3, 7, 11 at d = 1): the slope over k is the K loop, the intercept the staging + epilogue of a tile."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowhigh_amd import hip                 # noqa: E402
from flowhigh_amd import vocoder as V        # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = int(sys.argv[2]) if len(sys.argv) > 2 else 480000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
xs = [torch.randn(1, C, L, generator=g).to(dev) for _ in range(3)]
rs = [torch.randn(1, C, L, generator=g).to(dev) for _ in range(3)]
outs = [torch.empty(1, C, L, device=dev) for _ in range(3)]
bias = torch.zeros(C, device=dev)


def timed(fn, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(ks, d, direct):
    pack = V.pack_narrow_bf_weight if direct else V.pack_amp_weight
    us = [pack(torch.randn(C, C, k, generator=g) * (C * k) ** -0.5, C).to(dev) for k in ks]
    groups = [V.make_amp_group([V.make_amp_seg(xs[i], us[i], k, direct=direct)], bias, [rs[i]], outs[i], L, direct=direct)
              for i, k in enumerate(ks)]
    tiles = V.amp_tile_list([L] * 3, 1, d, direct=direct).to(dev)
    n_tiles = tiles.shape[0]
    desc = hip.to_device_struct_array(groups, dev)
    lib, st, vec = hip.lib(), hip.stream(), int(L % 4 == 0)
    if direct:
        fn = lambda: hip.check(lib.fh_narrow_conv_bf16x6_f32(desc.data_ptr(), 3, tiles.data_ptr(), n_tiles, C, d, vec, st), "narrow")
    else:
        fn = lambda: hip.check(lib.fh_amp_actconv_f32(desc.data_ptr(), 3, tiles.data_ptr(), n_tiles, C, d, V.amp_max_center(groups), vec | 2, st), "amp")
    return timed(fn)


print(f"C = {C}, L = {L}, three groups, bias + one residual; us per launch")
print(f"{'taps':>12} {'d':>2} {'winograd fp32':>14} {'direct bf16x6':>14}")
for ks, d in [((11, 7, 3), 1), ((11, 7, 3), 3), ((11, 7, 3), 5), ((1, 1, 1), 1), ((3, 3, 3), 1), ((7, 7, 7), 1), ((11, 11, 11), 1)]:
    a, b = run(ks, d, False), run(ks, d, True)
    print(f"{str(ks):>12} {d:>2} {a:14.1f} {b:14.1f}")
