"""Host-side split of one GraphPlan: python wall vs the C call's own phases (GCONV_COUNT_STAMPS)."""
import os, sys, time
os.environ["GCONV_COUNT_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
batch, _ = ga.synth.make_workload("hiv")
ei = batch.edge_index.to(dev); n = batch.num_nodes
for _ in range(5): ga.GraphPlan(ei, n, True)
torch.cuda.synchronize()
for _ in range(4):
    t0 = time.perf_counter(); ga.GraphPlan(ei, n, True); t1 = time.perf_counter()
    print(f"python wall {1e6 * (t1 - t0):.1f} us", file=sys.stderr)
