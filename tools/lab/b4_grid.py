"""B4 grid for small / mid graphs: one node block per wave slot (round 2) against four (round 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
os.environ["GCONV_NO_FUSED_GATHER"] = "1"
def ev(fn, iters=30, warm=5):
    best = 1e9
    for _ in range(3):
        for _ in range(warm): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best
ga.dist.pin_to_local_cores()
for n, d, r in ((2708, 32, 16), (19717, 32, 16), (8000, 128, 32), (32000, 128, 32), (65000, 128, 32), (20000, 300, 8)):
    b = ga.synth.arxiv_like(n, int(n * 3.53), d, 0)
    x = b.x.to(dev); ei = b.edge_index.to(dev)
    mod = ga.GConvAdapter(d, r, learnable_scalar=True).to(dev)
    with torch.no_grad():
        for p in mod.parameters(): p.normal_(0, .1)
    g = torch.randn(n, d, device=dev)
    xg = x.clone().requires_grad_(True)
    out = mod(xg, ei)
    leaves = [xg] + list(mod.parameters())
    def bw():
        for q in leaves: q.grad = None
        out.backward(g, retain_graph=True)
    print(f"per_slot={os.environ.get('GCONV_B4_BLOCKS_PER_SLOT','1')} N={n:6d} d={d:3d} r={r:2d}  bwd {ev(bw):6.1f} us", flush=True)
    ga.PLAN_CACHE.clear()
