import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
batch, cfg = ga.synth.make_workload(sys.argv[1] if len(sys.argv) > 1 else "hiv")
ei = batch.edge_index.to(dev); n = batch.num_nodes
for _ in range(5):
    ga.GraphPlan(ei, n, True)
torch.cuda.synchronize()
lib = ga._lib.load()
t0 = time.perf_counter()
for _ in range(200):
    lib.gconv_graph_build_bytes(ei.shape[1], n)
print("build_bytes: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter()
for _ in range(200):
    ga.GraphPlan(ei, n, True)
torch.cuda.synchronize()
print("GraphPlan  : %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
