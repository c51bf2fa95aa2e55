import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "esol"
batch, cfg = ga.synth.make_workload(name)
d, r = cfg["d"], cfg["r"]
x = batch.x.to(dev); ei = batch.edge_index.to(dev); g = torch.randn_like(x)
mods = [ga.GConvAdapter(d, r, learnable_scalar=True).to(dev) for _ in range(5)]
params = [p for m in mods for p in m.parameters()]
ga.get_plan(ei, x.shape[0], True)
acc = [0.0] * 4
N = 300
for it in range(N + 20):
    if it == 20:
        torch.cuda.synchronize(); acc = [0.0] * 4; T0 = time.perf_counter()
    t0 = time.perf_counter()
    for p in params: p.grad = None
    t1 = time.perf_counter()
    h = x
    for m in mods: h = m(h, ei)
    t2 = time.perf_counter()
    h.backward(g)
    t3 = time.perf_counter()
    del h
    t4 = time.perf_counter()
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3
torch.cuda.synchronize()
tot = (time.perf_counter() - T0) / N * 1e6
print(name, "grad=None %.0f | forward x5 %.0f | backward %.0f | del %.0f | total/step %.0f us" % tuple([a / N * 1e6 for a in acc] + [tot]))
