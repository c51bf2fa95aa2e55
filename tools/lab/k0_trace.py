"""`python k0_trace.py hiv` — 200 builds of one workload's GraphPlan (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "hiv"
batch, _ = ga.synth.make_workload(name)
ei = batch.edge_index.to(dev); n = batch.num_nodes
for _ in range(5): ga.GraphPlan(ei, n, True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): ga.GraphPlan(ei, n, True)
torch.cuda.synchronize()
print(f"{name}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per build")
# the floor: one empty-ish kernel + 256-byte D2H + synchronise
x = torch.zeros(64, dtype=torch.int32, device=dev); h = torch.empty(64, dtype=torch.int32).pin_memory()
for _ in range(5): x.add_(1); h.copy_(x, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): x.add_(1); h.copy_(x, non_blocking=True); torch.cuda.current_stream().synchronize()
print(f"floor (1 launch + D2H + sync): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
