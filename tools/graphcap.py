"""HIP-graph capture of a whole adapter step (static graph: the transductive setting): eager vs\nreplay time and bitwise equality of the gradients.  python tools/graphcap.py [cora|pubmed|esol]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cora"
batch, cfg = ga.synth.make_workload(name)
d, r = cfg["d"], cfg["r"]
x = batch.x.to(dev); ei = batch.edge_index.to(dev); gout = torch.randn_like(x)
mods = [ga.GConvAdapter(d, r, learnable_scalar=True).to(dev) for _ in range(cfg["adapters"])]
with torch.no_grad():
    for m in mods:
        for p in m.parameters(): p.normal_(0, 0.1)
params = [p for m in mods for p in m.parameters()]
ga.get_plan(ei, x.shape[-2], True)
def step():
    for p in params: p.grad = None
    h = x
    for m in mods: h = m(h, ei)
    h.backward(gout)
def timeit(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
ga.dist.pin_to_local_cores()
eager = timeit(step)
ref = [p.grad.clone() for p in params]
# capture
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
for p in params: p.grad = None
with torch.cuda.graph(g):
    h = x
    for m in mods: h = m(h, ei)
    h.backward(gout)
g.replay(); torch.cuda.synchronize()
ok = all(torch.equal(p.grad, r0) for p, r0 in zip(params, ref))
rep = timeit(g.replay)
print(f"{name}: eager {eager:.1f} us/step, graph replay {rep:.1f} us/step, grads identical: {ok}")
