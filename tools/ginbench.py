"""Device time of the frozen GIN layer's aggregation (csrc/gin.hip) and of the whole layer:
    python tools/ginbench.py [hiv|hiv4096|esol ...]
HIP events around back-to-back launches, best of 3. Algorithmic bytes of the aggregation:
read x + write agg (2 N d 4) + rowptr + col + the [N, 18] code counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga

dev = torch.device("cuda:0")


def ev(fn, iters=200, warm=10, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


def run(name):
    batch, cfg = ga.synth.make_workload(name)
    d, n = cfg["d"], batch.num_nodes
    ei = batch.edge_index.to(dev)
    e = ei.shape[1]
    g = torch.Generator().manual_seed(0)
    ea = torch.stack([torch.randint(0, 4, (e,), generator=g), torch.randint(0, 3, (e,), generator=g)], 1).to(dev)
    x = batch.x.to(dev)
    layer = ga.MolecularGINConv(d).to(dev)
    for p in layer.parameters():
        p.requires_grad_(False)
    plan = ga.get_plan(ei, n, False)
    counts = ga.gin.edge_code_counts(ei, ea, n)
    t, dr = layer.edge_embedding_type.weight, layer.edge_embedding_direction.weight
    agg_us = ev(lambda: ga.gin._aggregate(plan.fwd, x, counts, t, dr))
    bwd_us = ev(lambda: ga.gin._aggregate(plan.bwd, x, None, None, None))
    cnt_us = ev(lambda: ga._lib.load().gconv_gin_code_count(ei.data_ptr(), ea.data_ptr(), e, n, counts.data_ptr(),
                                                            torch.cuda.current_stream().cuda_stream))
    bytes_f = 2 * n * d * 4 + 4 * (n + 1) + 4 * plan.fwd.nnz + 4 * 18 * n
    bytes_b = 2 * n * d * 4 + 4 * (n + 1) + 4 * plan.fwd.nnz
    xg = x.clone().requires_grad_(True)
    gout = torch.randn_like(x)

    def layer_step():
        xg.grad = None
        layer(xg, ei, ea).backward(gout)

    def torch_step():   # the reference's op order with torch ops on the same device
        xg.grad = None
        loops = torch.arange(n, device=dev)
        ei2 = torch.cat([ei, torch.stack([loops, loops])], 1)
        sa = torch.zeros(n, 2, dtype=ea.dtype, device=dev); sa[:, 0] = 4
        at = torch.cat([ea, sa], 0)
        emb = t[at[:, 0]] + dr[at[:, 1]]
        msg = xg.index_select(0, ei2[0]) + emb
        agg = torch.zeros_like(xg).index_add_(0, ei2[1], msg)
        layer.mlp(agg).backward(gout)

    step_us, ref_us = ev(layer_step, iters=50), ev(torch_step, iters=50)
    print(f"{name}: N={n} E={e} d={d}  aggregate {agg_us:6.1f} us ({bytes_f / agg_us / 1e3:6.0f} GB/s)  "
          f"d-agg->dx {bwd_us:6.1f} us ({bytes_b / bwd_us / 1e3:6.0f} GB/s)  code counts {cnt_us:5.1f} us  |  "
          f"layer fwd+bwd {step_us:7.1f} us  (torch ops in the reference's order: {ref_us:7.1f} us)")


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["hiv", "hiv4096", "esol"]):
        run(w)
