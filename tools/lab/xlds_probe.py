"""Largest tile (rows incl. halo) of a workload's plan under the current tile-size knobs: does the tile's X / G copy
fit LDS beside the tile state (the round-4 XL experiment)?  python tools/lab/xlds_probe.py hiv4096"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["hiv4096"]:
    batch, cfg = ga.synth.make_workload(name)
    plan = ga.get_plan(batch.edge_index.to(dev), batch.num_nodes, True)
    g = plan.struct
    rec = plan.arena_view(g.tiles_fwd.tiles, 8 * plan.n_tiles, torch.int32).view(-1, 8).cpu()   # first, count, n1, n0, ...
    print(f"{name}: tiles {g.n_tiles} tile_nodes_max {g.tile_nodes_max} tile_edges_max {g.tile_edges_max} -> X copy "
          f"{g.tile_nodes_max * cfg['d'] * 4 / 1024:.0f} KB; own rows per tile max {int(rec[:, 1].max())} mean {rec[:, 1].float().mean():.1f}; "
          f"rows incl. halo max {int(rec[:, 3].max())}; halo rows in all {int((rec[:, 3] - rec[:, 1]).sum())} of {batch.num_nodes}")
