import torch, time
x = torch.empty(2000*49*1024, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
d = t(lambda: (x.fill_(1.0), y.fill_(2.0)))
print("fill 2 x 401 MB: %.1f us  %.2f TB/s" % (d*1e6, 2*x.numel()*4/d/1e12))
d = t(lambda: y.copy_(x))
print("copy 401 MB: %.1f us  %.2f TB/s (r+w)" % (d*1e6, 2*x.numel()*4/d/1e12))
d = t(lambda: x.sum())
print("sum 401 MB: %.1f us  %.2f TB/s" % (d*1e6, x.numel()*4/d/1e12))
