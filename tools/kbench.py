"""Per-kernel micro-benchmark through the C ABI (HIP events on the launch stream).
    python tools/kbench.py [arxiv|hiv|esol ...]
Prints average launch time and algorithmic GB/s of K1, K2, K3 and of the adapter fwd / bwd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga

dev = torch.device("cuda:0")


def ev(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def run(name, normalize=True):
    batch, cfg = ga.synth.make_workload(name)
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    x = batch.x.reshape(-1, d).to(dev)
    ei = batch.edge_index.to(dev)
    plan = ga.get_plan(ei, n, normalize)
    nnz = plan.fwd.nnz
    lib = ga._lib.load()
    fp = lib.gconv_padded_rank(r)
    wd = torch.randn(r, d, device=dev) * .1; wu = torch.randn(d, r, device=dev) * .1
    bu = torch.randn(d, device=dev); s = torch.ones(1, device=dev)
    h = torch.randn(n, fp, device=dev); y = torch.empty_like(h)
    ws = torch.empty(max(int(lib.gconv_spmm_workspace_bytes(plan.fwd.ref(), fp)), 16), dtype=torch.uint8, device=dev)
    fl = ga._lib.SPMM_IN_PRESCALED | ga._lib.SPMM_OUT_PRESCALE
    t1 = ev(lambda: ga.gemm_down(x, wd, plan.dis))
    t2 = ev(lambda: ga.spmm(plan, h, flags=fl, out=y, workspace=ws))
    t3 = ev(lambda: ga.gemm_up_fused(h, wu, bu, s, x, True))
    b1 = 4 * n * d + 4 * n * fp; b2 = 4 * (n + 1) + 4 * nnz + 4 * n + 8 * n * fp; b3 = 8 * n * d + 4 * n * fp
    print(f"{name}: N={n} nnz={nnz} d={d} r={r} | K1 down {t1:7.1f}us {b1/t1/1e3:7.0f}GB/s | "
          f"K2 spmm {t2:7.1f}us {b2/t2/1e3:7.0f}GB/s | K3 up {t3:7.1f}us {b3/t3/1e3:7.0f}GB/s")
    mod = ga.GConvAdapter(d, r, learnable_scalar=True, normalize=normalize).to(dev)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, .1)
    xg = x.clone().requires_grad_(True)
    g = torch.randn(n, d, device=dev)
    out = mod(xg, ei)
    leaves = [xg] + list(mod.parameters())
    tf = ev(lambda: mod(xg, ei), 20)
    def bw():
        for q in leaves: q.grad = None
        out.backward(g, retain_graph=True)
    tb = ev(bw, 20)
    print(f"    adapter fwd {tf:7.1f}us  bwd {tb:7.1f}us")


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["arxiv", "hiv"]):
        run(w)
