import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
n, d, r = 3001, 128, 32
g = torch.Generator().manual_seed(5)
u = torch.randint(n, (9000,), generator=g)
v = (n * torch.rand(9000, generator=g) ** 3).long().clamp_(max=n - 1)
hub = torch.stack([torch.randint(n, (700,), generator=g), torch.zeros(700, dtype=torch.long)])
pairs = torch.cat([torch.stack([u, v]), hub], dim=1)
ei = torch.cat([pairs, pairs.flip(0)], dim=1).to(dev)
x = torch.randn(n, d, generator=torch.Generator().manual_seed(1)).to(dev)
gr = torch.randn(n, d, generator=torch.Generator().manual_seed(2)).to(dev)
mod = ga.GConvAdapter(d, r, learnable_scalar=True, normalize=True, non_linearity="silu").to(dev)
with torch.no_grad():
    for p in mod.parameters():
        p.normal_(0, 0.1)
for mode in ("1", "0", "1", "0"):
    os.environ["GCONV_NO_FUSED_GATHER"] = mode
    vals = []
    for rep in range(6):
        if rep == 3:   # churn the allocator between repeats
            junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(8)]
            del junk
        mod.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        out = mod(xg, ei)
        out.backward(gr)
        torch.cuda.synchronize()
        vals.append((mod.scalar.grad.item(), mod.conv_up.bias.grad.double().sum().item(), xg.grad.double().sum().item()))
    print("no_fused=" + mode, ["%.6f" % v[0] for v in vals], "same" if len(set(vals)) == 1 else "DIFFER")
