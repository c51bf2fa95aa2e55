"""The gather-fused general path on uniform low-degree graphs (no long rows) of growing size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
ga.dist.pin_to_local_cores()
os.environ["GCONV_FUSED_GATHER_MAX_NODES"] = "100000000"
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
for d, r in ((32, 16), (128, 32), (300, 8)):
    for n in (2708, 8000, 20000, 40000, 65000, 130000):
        b = ga.synth.planetoid_like(n, int(n * 2.2), d, 0)
        x = b.x.reshape(-1, d).to(dev); ei = b.edge_index.to(dev)
        mod = ga.GConvAdapter(d, r, learnable_scalar=True).to(dev)
        with torch.no_grad():
            for p in mod.parameters(): p.normal_(0, .1)
        g = torch.randn(n, d, device=dev)
        res = {}
        for mode in ("separate", "fused"):
            os.environ["GCONV_NO_FUSED_GATHER"] = "1" if mode == "separate" else "0"
            xg = x.clone().requires_grad_(True)
            out = mod(xg, ei)
            leaves = [xg] + list(mod.parameters())
            def bw():
                for q in leaves: q.grad = None
                out.backward(g, retain_graph=True)
            res[mode] = (ev(lambda: mod(xg, ei)), ev(bw))
        plan = ga.get_plan(ei, n, True)
        print(f"d={d:3d} r={r:2d} N={n:6d} maxdeg={plan.struct.max_degree:3d} tiles={plan.n_tiles} | separate fwd {res['separate'][0]:6.1f} bwd {res['separate'][1]:6.1f} | fused fwd {res['fused'][0]:6.1f} bwd {res['fused'][1]:6.1f}", flush=True)
        ga.PLAN_CACHE.clear()
