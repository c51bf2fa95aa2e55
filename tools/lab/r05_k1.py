"""K1 (gemm_down) and B1-shaped calls at S-arxiv scale under the library named by GCONV_ADAPTER_LIB: best of 5 event
averages. Used by tools/lab/r05_k1.sh (lab variants of the ring schedule / store policy / grid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
def ev(fn, iters=100, warm=20, reps=5):
    best = 1e9
    for _ in range(reps):
        for _ in range(warm): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best
tag = os.environ.get("LAB_TAG", "product")
for n, d, r in ((170000, 128, 32), (170000, 128, 8), (102952, 300, 8)):
    x = torch.randn(n, d, device=dev)
    dis = torch.rand(n, device=dev)
    wd = torch.randn(r, d, device=dev) * .1
    t = ev(lambda: ga.gemm_down(x, wd, dis))
    fp = max(4, 1 << (r - 1).bit_length())
    by = 4 * n * d + 4 * n * fp
    print(f"{tag:24s} N={n} d={d} r={r:2d}: {t:6.2f} us  {by/t/1e3:6.0f} GB/s", flush=True)
