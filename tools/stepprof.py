"""Host-side phases of the bench step (fresh plan per step, prefetch thread), per workload:
    python tools/stepprof.py [esol|hiv] [fresh|cached] [heat] [nopin] [st]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "esol"
fresh = (sys.argv[2] != "cached") if len(sys.argv) > 2 else True
if "heat" in sys.argv:   # what a preceding heavy phase (S-arxiv measurement) does to the step
    bench.measure_arxiv(dev, banded_control=False)
stream = bench.BatchStream(name, 0, dev, 4, fresh)
pinned = [] if "nopin" in sys.argv else ga.dist.pin_to_local_cores()   # as bench.py does
cfg = stream.cfg
mods = [ga.GConvAdapter(cfg["d"], cfg["r"], learnable_scalar=True).to(dev) for _ in range(cfg["adapters"])]
params = [p for m in mods for p in m.parameters()]
if "st" in sys.argv:     # autograd nodes on the calling thread (no hand-off to the device thread)
    torch.autograd.set_multithreading_enabled(False)
N = 300
acc = [0.0] * 6
for it in range(N + 30):
    if it == 30:
        torch.cuda.synchronize(); acc = [0.0] * 6; T0 = time.perf_counter()
    t0 = time.perf_counter()
    batch, x, ei, gout = stream.next(True)
    t1 = time.perf_counter()
    for p in params: p.grad = None
    t2 = time.perf_counter()
    h = mods[0](x, ei)
    t3 = time.perf_counter()
    for m in mods[1:]: h = m(h, ei)
    t4 = time.perf_counter()
    h.backward(gout)
    t5 = time.perf_counter()
    if fresh: ga.PLAN_CACHE.discard(ei, x.shape[-2], True)
    del h
    t6 = time.perf_counter()
    for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))): acc[i] += b - a
torch.cuda.synchronize()
tot = (time.perf_counter() - T0) / N * 1e6
print(name, "fresh" if fresh else "cached", "pinned" if pinned else "not pinned", "| next+prefetch %.0f | grad=None %.0f | fwd[0] (plan wait) %.0f | fwd[1:] %.0f | backward %.0f | discard %.0f | total/step %.0f us"
      % tuple([a / N * 1e6 for a in acc] + [tot]))
