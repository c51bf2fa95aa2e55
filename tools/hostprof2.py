"""Where does the host time of a step go? cProfile by tottime + direct timing of the C calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "hiv"
batch, cfg = ga.synth.make_workload(name)
d, r = cfg["d"], cfg["r"]
x = batch.x.reshape(-1, d).to(dev); ei = batch.edge_index.to(dev)
mods = [ga.GConvAdapter(d, r, learnable_scalar=True, normalize=True).to(dev) for _ in range(cfg["adapters"])]
g = torch.randn_like(x)
params = [p for m in mods for p in m.parameters()]

def fwd_only():
    h = x
    for m in mods:
        h = m(h, ei)
    return h

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h = (time.perf_counter() - t0) / n; torch.cuda.synchronize()
    return h * 1e6, (time.perf_counter() - t0) / n * 1e6

ga.get_plan(ei, x.shape[0], True)
print("5x forward (cached plan)  host/total: %.0f / %.0f us" % timeit(fwd_only))
def fb():
    for p in params: p.grad = None
    fwd_only().backward(g)
print("5x fwd+bwd (cached plan)  host/total: %.0f / %.0f us" % timeit(fb))
def prep():
    ga.PLAN_CACHE.clear(); ga.get_plan(ei, x.shape[0], True)
print("prep                      host/total: %.0f / %.0f us" % timeit(prep))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fb()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
