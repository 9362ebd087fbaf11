"""HBM write / read / copy rates of plain torch kernels at several sizes (is the write path the ceiling of the write-heavy blocks?)."""
import torch
d = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=d); y = torch.empty_like(x)
    x.normal_()
    tw = t(lambda: y.zero_()); tf = t(lambda: y.fill_(1.5))
    tr = t(lambda: x.sum()) ; tc = t(lambda: y.copy_(x)); ta = t(lambda: torch.relu_(y))
    print("%5d MB: zero_ %.2f TB/s  fill_ %.2f TB/s | sum (read) %.2f TB/s | copy %.2f TB/s each way (%.2f total) | relu_ in place %.2f TB/s each way"
          % (mb, mb / 1e6 * 1.048576 / tw, mb / 1e6 * 1.048576 / tf, mb / 1e6 * 1.048576 / tr, mb / 1e6 * 1.048576 / tc, 2 * mb / 1e6 * 1.048576 / tc, mb / 1e6 * 1.048576 / ta))
    del x, y
