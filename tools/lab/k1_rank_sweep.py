"""K1 (gemm_down) time at S-arxiv scale as a function of the rank (MFMA work halves with it, the
[N, d] stream does not) — separates the MFMA-bound share from the memory-bound share."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
def ev(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for n, d in ((170000, 128), (102952, 300), (170000, 64), (170000, 256)):
    x = torch.randn(n, d, device=dev)
    dis = torch.rand(n, device=dev)
    for r in (4, 8, 16, 32, 64):
        wd = torch.randn(r, d, device=dev) * .1
        t = ev(lambda: ga.gemm_down(x, wd, dis))
        fp = max(4, 1 << (r - 1).bit_length())
        by = 4 * n * d + 4 * n * fp
        print(f"N={n} d={d} r={r:2d}: {t:6.1f} us  {by/t/1e3:6.0f} GB/s  {2*n*d*max(fp,16)/t/1e6:6.1f} TFLOP/s(issued)", flush=True)
