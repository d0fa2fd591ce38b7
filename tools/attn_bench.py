"""fh_attention_f32 (16 heads x 64, fp32) at the transformer's shapes.  python tools/attn_bench.py"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip
DEV = torch.device("cuda:0")
for B, N in ((1, 50), (1, 1000), (8, 1000), (32, 1000), (1, 3000), (8, 3000)):
    qkv = torch.randn(B * N, 3072, device=DEV) * 0.3
    out = torch.empty(B * N, 1024, device=DEV)
    run = lambda: hip.check(hip.lib().fh_attention_f32(qkv.data_ptr(), out.data_ptr(), B, N, 16, 10.0, hip.stream()), "attn")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"B={B:2d} N={N:5d}: {us:9.1f} us  {4.0 * B * 16 * N * N * 64 / us / 1e6:6.1f} TFLOP/s")
