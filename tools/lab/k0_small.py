"""Graph-build (K0) time for small batches, in-stream (what a drop-in user without prefetch_plan pays per batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
for name, graphs in (("esol", None), ("hiv", 32), ("hiv", 128), ("hiv", None), ("cora", None)):
    batch, cfg = ga.synth.make_workload(name, graphs=graphs)
    ei = batch.edge_index.to(dev)
    n = batch.num_nodes
    for _ in range(5):
        ga.GraphPlan(ei, n, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ga.GraphPlan(ei, n, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"{name}:{graphs} n={n} e={ei.shape[1]}  build {dt*1e6:7.1f} us wall per plan", flush=True)
