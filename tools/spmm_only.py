"""Runs only the S-arxiv normalized SpMM (rank 32, production flags) a few times — target of the
rocprofv3 --pmc passes that give the roofline `traffic` figure."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
if "--banded" in sys.argv:   # the locality control of bench.py --banded (1 M nodes, neighbours within +-2048)
    batch, cfg = ga.synth.banded_like(1_000_000, 8, 2048), {"r": 32}
else:
    batch, cfg = ga.synth.make_workload("arxiv")
n, r = batch.num_nodes, cfg["r"]
plan = ga.get_plan(batch.edge_index.to(dev), n, True)
h = torch.randn(n, r, device=dev); y = torch.empty_like(h)
lib = ga._lib.load()
ws = torch.empty(max(int(lib.gconv_spmm_workspace_bytes(plan.fwd.ref(), r)), 16), dtype=torch.uint8, device=dev)
fl = ga._lib.SPMM_IN_PRESCALED | ga._lib.SPMM_OUT_PRESCALE
for _ in range(20):
    ga.spmm(plan, h, flags=fl, out=y, workspace=ws)
torch.cuda.synchronize()
print("nnz", plan.fwd.nnz, "n_long", plan.fwd.n_long, "n_seg", plan.fwd.n_seg)
if "--time" in sys.argv:   # launch time, HIP events, best of 5 x 50 (tools/lab/spmm_policy.sh)
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            ga.spmm(plan, h, flags=fl, out=y, workspace=ws)
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / 50 * 1e3)
    print("spmm us %.2f  lib %s" % (best, os.path.basename(os.environ.get("GCONV_ADAPTER_LIB", "product"))))
