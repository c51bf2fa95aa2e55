import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
for name in ("cora", "pubmed"):
    batch, cfg = ga.synth.make_workload(name)
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    ei = batch.edge_index.to(dev)
    mod = ga.GConvAdapter(d, r, learnable_scalar=True).to(dev)
    with torch.no_grad():
        for p in mod.parameters(): p.normal_(0, .1)
    x = batch.x.reshape(1, n, d).to(dev).requires_grad_(True)
    g = torch.randn(1, n, d, device=dev)
    out = mod(x, ei)
    leaves = [x] + list(mod.parameters())
    def bwd():
        for q in leaves: q.grad = None
        out.backward(g, retain_graph=True)
    for _ in range(10): bwd()
    torch.cuda.synchronize()
    hs = []
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        t0 = time.perf_counter(); bwd(); hs.append(time.perf_counter() - t0)
    b.record(); b.synchronize()
    hs.sort()
    print(name, "bwd device-span/iter %.1f us  host median %.1f us  host max %.1f us" % (a.elapsed_time(b) * 20, hs[25] * 1e6, hs[-1] * 1e6), flush=True)
    plan = ga.get_plan(ei, n, True)
    print("  n_long", plan.fwd.n_long, "n_seg", plan.fwd.n_seg, "n_tiles", plan.n_tiles, "sym", plan.symmetric)
