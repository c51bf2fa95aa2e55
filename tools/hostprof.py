"""Host-side cost breakdown of one adapter step (prep / forward call / backward call)."""
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


def timed(fn, n=200, sync_each=False):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n
    return host * 1e6, total * 1e6


def prep():
    ga.PLAN_CACHE.clear(); ga.get_plan(ei, x.shape[0], True)


print(name, "nodes", x.shape[0])
print("prep (host, incl. its syncs)      : %.1f us" % timed(prep)[0])
plan = ga.get_plan(ei, x.shape[0], True)
xg = x.clone().requires_grad_(True)
print("fwd call, 1 adapter (host / total): %.1f / %.1f us" % timed(lambda: mods[0](xg, ei)))
out = mods[0](xg, ei)
leaves = [xg] + list(mods[0].parameters())


def bwd():   # (the parameters are not graph inputs under direct gradient deposit: plain backward)
    for q in leaves:
        q.grad = None
    out.backward(g, retain_graph=True)


print("bwd call, 1 adapter (host / total): %.1f / %.1f us" % timed(bwd))


def step():
    ga.PLAN_CACHE.clear()
    h = x
    for m in mods:
        h = m(h, ei)
    for p_ in (p for m in mods for p in m.parameters()):
        p_.grad = None
    h.backward(g)


print("full step (host / total)          : %.1f / %.1f us" % timed(step, 100))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
