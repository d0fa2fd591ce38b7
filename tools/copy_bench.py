"""Device copy bandwidth at the activation kernel's sizes (what a read-once / write-once pass can reach)."""
import torch
DEV = torch.device('cuda:0')
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for n in (3 * 768 * 5000, 3 * 384 * 20000, 3 * 192 * 60000, 4 * 3 * 192 * 60000, 16 * 3 * 192 * 60000):
    x = torch.randn(n, device=DEV); y = torch.empty_like(x)
    t = bench(lambda: y.copy_(x))
    t2 = bench(lambda: torch.add(x, 1.0, out=y))
    print(f"n = {n:10d} ({n * 4 / 1e6:7.1f} MB): copy_ {t:7.1f} us {n * 8 / t / 1e6:5.2f} TB/s   add {t2:7.1f} us {n * 8 / t2 / 1e6:5.2f} TB/s")
