"""Wall time per GraphPlan (in-stream, synchronous) for the bench's graphs + kernel list of one build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
for name, graphs in (("esol", None), ("hiv", None), ("hiv4096", None), ("arxiv", None)):
    batch, _ = ga.synth.make_workload(name, graphs=graphs)
    ei = batch.edge_index.to(dev); n = batch.num_nodes
    for _ in range(5): ga.GraphPlan(ei, n, True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): ga.GraphPlan(ei, n, True)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 30)
    bg = 1e9
    for _ in range(3):   # the same build the way the prefetch worker asks for it (GCONV_BUILD_BACKGROUND; GCONV_BG_BUILD selects)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): ga.GraphPlan(ei, n, True, background=True)
        torch.cuda.synchronize(); bg = min(bg, (time.perf_counter() - t0) / 30)
    p = ga.GraphPlan(ei, n, True)
    print(f"{name}: n={n} tiles={p.n_tiles} build wall {best * 1e6:.1f} us; background flavour {bg * 1e6:.1f} us "
          f"(GCONV_BG_BUILD={os.environ.get('GCONV_BG_BUILD', 'default')}, cooperative={os.environ.get('GCONV_COUNT_COOPERATIVE', '0')})")
