"""One S-arxiv adapter (d = 128, rank 32, normalized), forward + backward x 10: the program behind the
`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES` pass of tools/pmc_mfma.sh (MFMA utilisation
of the projection kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
batch, cfg = ga.synth.make_workload("arxiv")
mod = ga.GConvAdapter(cfg["d"], cfg["r"], learnable_scalar=True, normalize=True).to(dev)
with torch.no_grad():
    for p in mod.parameters():
        p.normal_(0, 0.1)
x = batch.x.to(dev).requires_grad_(True)
ei = batch.edge_index.to(dev)
g = torch.randn_like(x)
for _ in range(10):
    mod.zero_grad(set_to_none=True)
    x.grad = None
    mod(x, ei).backward(g)
torch.cuda.synchronize()
