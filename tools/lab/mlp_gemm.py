"""fp32 GEMM rates of the frozen GIN MLP shapes through torch (rocBLAS / hipBLASLt)."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
def ev(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3
print("preferred blas:", torch.backends.cuda.preferred_blas_library())
for n in (13104, 102952):
    x = torch.randn(n, 300, device=dev); w1 = torch.randn(600, 300, device=dev); b1 = torch.randn(600, device=dev)
    h = torch.randn(n, 600, device=dev); w2 = torch.randn(300, 600, device=dev); b2 = torch.randn(300, device=dev)
    for name, fn, fl in (("mm   x@W1^T", lambda: torch.mm(x, w1.t()), 2 * n * 300 * 600),
                         ("addmm b1", lambda: torch.addmm(b1, x, w1.t()), 2 * n * 300 * 600),
                         ("mm   h@W2^T", lambda: torch.mm(h, w2.t()), 2 * n * 300 * 600),
                         ("mm   g@W2 (bwd)", lambda: torch.mm(x, w2), 2 * n * 300 * 600)):
        t = ev(fn)
        print(f"N={n:6d} {name:18s} {t:8.1f} us  {fl/t/1e6:6.1f} TFLOP/s", flush=True)
