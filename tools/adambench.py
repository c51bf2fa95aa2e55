"""FusedAdam vs torch.optim.Adam on the HIV trainable set (5 adapters x 5 tensors + head + 5 BN):
host time and device time of one optimizer step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")

def params():
    shapes = []
    for _ in range(5):
        shapes += [(8, 300), (8,), (300, 8), (300,), (1,)]      # adapter
        shapes += [(300,), (300,)]                                # BatchNorm affine
    shapes += [(1, 300), (1,)]                                    # graph_pred_linear
    return [torch.nn.Parameter(torch.randn(s, device=dev) * 0.1) for s in shapes]

def bench(opt, ps, n=300):
    for p in ps: p.grad = torch.randn_like(p)
    for _ in range(20): opt.step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(n): opt.step()
    b.record(); host = (time.perf_counter() - t0) / n * 1e6
    b.synchronize()
    return host, a.elapsed_time(b) / n * 1e3

ps = params(); print("trainable floats:", sum(p.numel() for p in ps), "in", len(ps), "tensors")
print("torch.optim.Adam (foreach) host %.1f us, stream time %.1f us per step" % bench(torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-5), ps))
ps = params()
print("torch.optim.Adam (fused=True) host %.1f us, stream time %.1f us per step" % bench(torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-5, fused=True), ps))
ps = params()
print("gconv FusedAdam           host %.1f us, stream time %.1f us per step" % bench(ga.FusedAdam(ps, lr=1e-3, weight_decay=1e-5), ps))
