"""Practical HBM ceilings on this box for map-sized buffers (torch kernels as the yardstick)."""
import torch, time
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (630, 1260):
    n = mb * 1024 * 1024 // 2
    xs = [torch.randn(n, device=dev, dtype=torch.bfloat16) for _ in range(3)]   # rotate: defeat the 256 MB infinity cache
    ys = [torch.empty_like(x) for x in xs]
    i = [0]
    def rd():
        i[0] = (i[0] + 1) % 3; return xs[i[0]].view(torch.int16).max()
    def cp():
        i[0] = (i[0] + 1) % 3; ys[i[0]].copy_(xs[i[0]])
    def fl():
        i[0] = (i[0] + 1) % 3; ys[i[0]].zero_()
    s = t(rd); print(f"{mb} MB read-only (max): {n*2/s/1e12:.2f} TB/s  {s*1e6:.0f} us")
    s = t(cp); print(f"{mb} MB copy (r+w):     {2*n*2/s/1e12:.2f} TB/s  {s*1e6:.0f} us")
    s = t(fl); print(f"{mb} MB fill (write):   {n*2/s/1e12:.2f} TB/s  {s*1e6:.0f} us")
