"""Phase times inside counting_build_kernel (GCONV_COUNT_STAMPS=1): odd intervals are barriers."""
import os, sys
os.environ["GCONV_COUNT_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["hiv"]:
    batch, _ = ga.synth.make_workload(name)
    ei = batch.edge_index.to(dev); n = batch.num_nodes
    for _ in range(6): ga.GraphPlan(ei, n, True)
    p = ga.GraphPlan(ei, n, True)
    t = p.tiles(False).cpu() if hasattr(p, "tiles") else None
    if t is not None:
        import numpy as np
        t = t.numpy()
        print("tile fields", ga.graph.TILE_FIELDS if hasattr(ga.graph, "TILE_FIELDS") else "")
        order = np.argsort(-(t[:, 3] - t[:, 1]))[:8]
        print("largest halos (record rows):"); print(t[order])
