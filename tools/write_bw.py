"""Diagnostic (not product): streaming write / read / copy bandwidth of the device as torch sees it (fill_, sum, copy_ of 4 GiB)."""
import torch, time
x = torch.empty(1 << 29, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
gb = x.numel() * 8 / 1e9
print("fill  %.2f TB/s" % (gb / t(lambda: x.fill_(1.0)) / 1e3))
print("sum   %.2f TB/s" % (gb / t(lambda: x.sum()) / 1e3))
print("copy  %.2f TB/s (read + write)" % (2 * gb / t(lambda: y.copy_(x)) / 1e3))
