"""A/B of the gather-fused general path (fused_gather.hip) against the separate launches, through the
module: bit-exactness of out / saved state and device time of forward / backward.
    python tools/fusebench.py [arxiv cora pubmed ...]      env GCONV_UPG_WGPC / GCONV_BWG_WGPC = workgroups per CU"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga

dev = torch.device("cuda:0")
os.environ.setdefault("GCONV_FUSED_GATHER_MAX_NODES", "100000000")   # force the fused path in "fused" mode


def ev(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def run(name):
    batch, cfg = ga.synth.make_workload(name)
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    x = batch.x.reshape(-1, d).to(dev)
    ei = batch.edge_index.to(dev)
    mod = ga.GConvAdapter(d, r, learnable_scalar=True, normalize=True).to(dev)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, .1)
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
        bw(); torch.cuda.synchronize()
        grads = [q.grad.clone() for q in leaves]
        tf = ev(lambda: mod(xg, ei), 30)
        tb = ev(bw, 30)
        res[mode] = (out.detach().clone(), grads, tf, tb)
    same_out = torch.equal(res["separate"][0], res["fused"][0])
    same_g = all(torch.equal(a, b) for a, b in zip(res["separate"][1], res["fused"][1]))
    md = max((a - b).abs().max().item() for a, b in zip(res["separate"][1], res["fused"][1]))
    print(f"{name}: N={n} d={d} r={r}  separate fwd {res['separate'][2]:7.1f} bwd {res['separate'][3]:7.1f} us | "
          f"fused fwd {res['fused'][2]:7.1f} bwd {res['fused'][3]:7.1f} us | out bit-equal {same_out}, grads bit-equal {same_g} (max diff {md:.2e})",
          flush=True)


if __name__ == "__main__":
    for wname in (sys.argv[1:] or ["arxiv", "cora", "pubmed"]):
        run(wname)
