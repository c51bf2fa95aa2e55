import sys, torch
sys.path.insert(0, '/root/repo')
import gconv_adapter_amd as ga
dev = torch.device('cuda:0')
for name, graphs in (("esol", None), ("hiv", 128), ("hiv", 512)):
    batch, cfg = ga.synth.make_workload(name, graphs=graphs)
    d, r = cfg["d"], cfg["r"]
    mod = ga.GConvAdapter(d, r, learnable_scalar=True, normalize=True).to(dev)
    with torch.no_grad():
        for p in mod.parameters(): p.normal_(0, 0.1)
    x = batch.x.to(dev).requires_grad_(True); ei = batch.edge_index.to(dev)
    g = torch.randn_like(x)
    outs = []
    for it in range(6):
        mod.zero_grad(set_to_none=True); x.grad = None
        out = mod(x, ei); out.backward(g)
        outs.append([out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in mod.parameters()])
    names = ["out", "dx"] + [k for k, _ in mod.named_parameters()]
    for it in range(1, 6):
        for nm, a, b in zip(names, outs[0], outs[it]):
            if not torch.equal(a, b):
                print(name, graphs, "iter", it, nm, "differs: max abs", (a - b).abs().max().item())
    print(name, graphs, "checked")
