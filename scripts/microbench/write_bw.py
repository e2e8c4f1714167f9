"""HBM write / copy rates of torch's own kernels on 1 GiB fp32 buffers (the ceiling the GEMM epilogues are compared with)."""
import time
import torch
x = torch.empty(1024, 262144, device="cuda")
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
gb = x.numel() * 4 / 1e9
d = t(lambda: x.fill_(1.0)); print("fill_  1 GiB: %.1f us  %.2f TB/s written" % (d * 1e6, gb / d / 1e3))
d = t(lambda: y.copy_(x)); print("copy_  1 GiB: %.1f us  %.2f TB/s read + %.2f TB/s written" % (d * 1e6, gb / d / 1e3, gb / d / 1e3))
d = t(lambda: torch.mul(x, 2.0, out=y)); print("mul    1 GiB: %.1f us  %.2f TB/s each way" % (d * 1e6, gb / d / 1e3))
d = t(lambda: x.sum()); print("sum    1 GiB: %.1f us  %.2f TB/s read" % (d * 1e6, gb / d / 1e3))
