"""Micro-benchmark of fh_conv_grouped_f32 on synthetic shapes: isolates per-occupancy efficiency."""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip
from flowhigh_amd import vocoder as V

def run(c, L, ks, tile_cfg, B=1, reps=5, res=False, label=""):
    dev = 'cuda'
    bm = hip.lib().fh_conv_tile_m(tile_cfg)
    cpad = -(-c // bm) * bm
    ck = V.pick_ck(c)
    groups, keep = [], []
    flops = 0
    for k in ks:
        x = torch.randn(B, c, L, device=dev); out = torch.empty(B, c, L, device=dev)
        w = V.pack_conv_weight(torch.randn(c, c, k) * 0.02, cpad, ck).to(dev)
        b = torch.randn(c, device=dev)
        r = [torch.randn(B, c, L, device=dev)] if res else []
        keep += [x, out, w, b] + r
        offs = [t - (k - 1) // 2 for t in range(k)]
        groups.append(V.make_conv_group([V.make_conv_seg(x, w, c, offs)], b, r, out, c, cpad, L, L, L))
        flops += 2.0 * c * c * k * L * B
    d = hip.to_device_struct_array(groups, dev)
    lib, st = hip.lib(), hip.stream()
    bn = lib.fh_conv_tile_n(tile_cfg)
    nblocks = len(ks) * B * (cpad // bm) * -(-L // bn)
    for _ in range(2):
        hip.check(lib.fh_conv_grouped_f32(d.data_ptr(), len(groups), B, cpad, L, tile_cfg, ck, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.check(lib.fh_conv_grouped_f32(d.data_ptr(), len(groups), B, cpad, L, tile_cfg, ck, st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:34s} C={c} L={L} ks={ks} cfg={tile_cfg} blocks={nblocks:5d} {ms*1e3:8.1f} us  {flops/ms/1e9:6.1f} TF/s", flush=True)

if __name__ == "__main__" and len(sys.argv) < 2:
    # equal-size blocks, cfg0 (3 blocks/CU resident): 256 / 512 / 768 / 1536 / 3072 blocks
    for nb, lab in [(256, "1 block/CU"), (512, "2 blocks/CU"), (768, "3 blocks/CU"), (1536, "2 rounds of 3"), (3072, "4 rounds"), (6144, "8 rounds")]:
        L = nb // 6 * 128
        run(768, L, [7], 0, label=lab)
    run(768, 5000, [11, 7, 3], 0, label="stage0 conv1-like")
    run(768, 5000, [11, 7, 3], 0, res=True, label="stage0 conv2-like (+res)")
    run(768, 5000, [11, 7, 3], 5, label="stage0 conv, 128x64 tile")
    run(768, 5000, [7, 7, 7], 0, label="stage0 equal groups")
    run(384, 20000, [11, 7, 3], 0, label="stage1 conv")
    run(192, 60000, [11, 7, 3], 1, label="stage2 conv (192x128)")
    run(192, 60000, [11, 7, 3], 2, label="stage2 conv (96x256)")
    run(192, 60000, [11, 7, 3], 0, label="stage2 conv (128x128 padded)")
    run(96, 120000, [11, 7, 3], 2, label="stage3 conv")
    run(48, 240000, [11, 7, 3], 3, label="stage4 conv")
    run(24, 480000, [11, 7, 3], 4, label="stage5 conv")
