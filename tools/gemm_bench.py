"""fh_gemm_f32 and fh_gemm_bf16x6_f32 at the transformer's shapes (M = frames x batch).  python tools/gemm_bench.py"""
import sys, torch
sys.path.insert(0, '.')
from flowhigh_amd import hip
from flowhigh_amd.packing import pack_gemm_bf_weight
DEV = torch.device("cuda:0")
for M in (1000, 4000, 24000):
    for N, K, name in ((3072, 1024, "qkv"), (1024, 1024, "to_out"), (5460, 1024, "ff1 (geglu)"), (1024, 2730 + 22, "ff2"), (256, 1024, "to_pred")):
        Kp = -(-K // 32) * 32
        Np = -(-N // 128) * 128
        A = torch.randn(M, Kp, device=DEV) * 0.1
        W = torch.randn(Np, Kp, device=DEV) * 0.1
        geglu = name.startswith("ff1")
        if geglu:
            Np = -(-5504 // 128) * 128
            W = torch.randn(Np, Kp, device=DEV) * 0.1
        C = torch.empty(M, (5504 // 2) if geglu else N, device=DEV)
        Wb = pack_gemm_bf_weight(W.cpu()).to(DEV)
        res = []
        for bf in (False, True):
            run = lambda: hip.gemm(A, Wb if bf else W, C, M, 5504 if geglu else N, Kp, epilogue=hip.EPI_GEGLU if geglu else hip.EPI_LINEAR, bf=bf)
            for _ in range(5): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 50)
        fl = 2.0 * M * (5504 if geglu else N) * Kp
        print(f"M={M:6d} {name:12s} N={N:5d} K={Kp:5d}: fp32 {res[0]:8.1f} us {fl / res[0] / 1e6:6.1f} TFLOP/s | bf16 x 6 {res[1]:8.1f} us {fl / res[1] / 1e6:6.1f} TFLOP/s (fp32-equivalent)")
