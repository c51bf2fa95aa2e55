"""Synthetic graph workloads of the benchmark shapes.

The reference's datasets (MoleculeNet ESOL/HIV via
``scripts/finetune_inductive_learning.py:263-293``, Planetoid Cora/PubMed via
``src/dataset/transductive/dataset.py:405-429``, ogbn-arxiv) need the network
and PyG; these generators produce graphs of the same *shape* (node/edge counts,
degree structure, batching into one block-diagonal ``edge_index``) with pure
torch on the CPU and a fixed seed, so the identical tensors can feed the HIP
path and the CPU oracle.

All generators return ``edge_index`` as int64 ``[2, E]`` with row 0 = source
and row 1 = target (PyG convention), both directions of every bond emitted.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch


@dataclass
class GraphBatch:
    """One collated batch: what PyG's DataLoader hands to ``model(x, edge_index, ...)``."""
    x: torch.Tensor            # [N, d] fp32
    edge_index: torch.Tensor   # [2, E] int64
    graph_ptr: torch.Tensor    # [B+1] int64 node offsets per graph (host-side bookkeeping)
    num_graphs: int

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[-2])


def _molecule_sizes(kind: str, num_graphs: int, gen: torch.Generator) -> torch.Tensor:
    if kind == "esol":
        n = torch.randn(num_graphs, generator=gen) * 6.8 + 13.3
        return n.round().clamp_(2, 55).long()
    if kind == "hiv":
        n = torch.exp(torch.randn(num_graphs, generator=gen) * 0.42 + 3.14)
        return n.round().clamp_(2, 222).long()
    raise ValueError(f"unknown molecule kind {kind!r}")


def molecule_edges(sizes: torch.Tensor, gen: torch.Generator) -> Tuple[torch.Tensor, torch.Tensor]:
    """Block-diagonal molecule-like batch.

    Per graph: random tree (node k attaches to a uniform parent in
    ``[max(0, k-3), k-1]``) plus ``round(0.08 * n)`` ring-closing chords; both
    directions emitted, no self-loops. Returns (edge_index, graph_ptr)."""
    sizes = sizes.long()
    ptr = torch.zeros(sizes.numel() + 1, dtype=torch.long)
    ptr[1:] = sizes.cumsum(0)
    total = int(ptr[-1])
    gid = torch.repeat_interleave(torch.arange(sizes.numel()), sizes)
    local = torch.arange(total) - ptr[gid]
    child = torch.nonzero(local > 0).squeeze(1)
    k = local[child]
    span = torch.clamp(k, max=3)
    back = 1 + (torch.rand(child.numel(), generator=gen) * span).floor().long().clamp_(max=2)
    back = torch.minimum(back, k)
    parent = child - back
    # chords
    n_ch = (0.08 * sizes.double()).round().long()
    cg = torch.repeat_interleave(torch.arange(sizes.numel()), n_ch)
    csz = sizes[cg]
    a = (torch.rand(cg.numel(), generator=gen) * csz).floor().long()
    a = torch.minimum(a, csz - 1)
    off = 1 + (torch.rand(cg.numel(), generator=gen) * (csz - 1)).floor().long()
    off = torch.minimum(off, csz - 1)
    b = (a + off) % csz
    ca, cb = a + ptr[cg], b + ptr[cg]
    u = torch.cat([child, ca])
    v = torch.cat([parent, cb])
    edge_index = torch.stack([torch.cat([u, v]), torch.cat([v, u])])
    return edge_index, ptr


def molecule_batch(kind: str, num_graphs: int, hidden: int = 300, seed: int = 0) -> GraphBatch:
    gen = torch.Generator().manual_seed(seed)
    sizes = _molecule_sizes(kind, num_graphs, gen)
    ei, ptr = molecule_edges(sizes, gen)
    x = torch.randn(int(ptr[-1]), hidden, generator=gen)
    return GraphBatch(x=x, edge_index=ei, graph_ptr=ptr, num_graphs=num_graphs)


def planetoid_like(num_nodes: int, num_pairs: int, hidden: int = 32, seed: int = 0,
                   self_loops: bool = True) -> GraphBatch:
    """Shape-only stand-in for Cora (2708, 5278) / PubMed (19717, 44324):
    uniform undirected pairs, symmetrised, de-duplicated, self-loops pre-added the
    way ``scripts/finetune_transductive_learning.py:111-114`` builds ``adjs[0]``;
    ``x`` is ``[1, N, d]`` as NodeFormer passes it (``nodeformer.py:391``)."""
    gen = torch.Generator().manual_seed(seed)
    u = torch.randint(num_nodes, (num_pairs,), generator=gen)
    v = torch.randint(num_nodes, (num_pairs,), generator=gen)
    keep = u != v
    u, v = u[keep], v[keep]
    key = torch.unique(torch.cat([u * num_nodes + v, v * num_nodes + u]))
    src, dst = key // num_nodes, key % num_nodes
    if self_loops:
        loops = torch.arange(num_nodes)
        src, dst = torch.cat([src, loops]), torch.cat([dst, loops])
    x = torch.randn(1, num_nodes, hidden, generator=gen)
    return GraphBatch(x=x, edge_index=torch.stack([src, dst]),
                      graph_ptr=torch.tensor([0, num_nodes]), num_graphs=1)


def arxiv_like(num_nodes: int = 170_000, num_pairs: int = 600_000, hidden: int = 128,
               seed: int = 0) -> GraphBatch:
    """S-arxiv: ``u = randint(N)``, ``v = floor(N * rand()^3)`` (power-law-ish hubs at
    low index), ``u == v`` dropped, symmetrised (duplicates kept)."""
    gen = torch.Generator().manual_seed(seed)
    u = torch.randint(num_nodes, (num_pairs,), generator=gen)
    v = (num_nodes * torch.rand(num_pairs, generator=gen, dtype=torch.float64) ** 3).floor().long()
    v.clamp_(max=num_nodes - 1)
    keep = u != v
    u, v = u[keep], v[keep]
    ei = torch.stack([torch.cat([u, v]), torch.cat([v, u])])
    x = torch.randn(num_nodes, hidden, generator=gen)
    return GraphBatch(x=x, edge_index=ei, graph_ptr=torch.tensor([0, num_nodes]), num_graphs=1)


def banded_like(num_nodes: int = 1_000_000, degree: int = 8, band: int = 2048, hidden: int = 32,
                seed: int = 0, with_x: bool = False) -> GraphBatch:
    """Locality control for the SpMM roofline: every node links to ``degree / 2`` random
    nodes within +-``band`` positions (symmetrised), i.e. the same average degree as S-arxiv
    but with neighbours that sit in nearby cache lines — what node orderings of real
    citation / molecule graphs look like, as opposed to the uniformly random S-arxiv."""
    gen = torch.Generator().manual_seed(seed)
    half = degree // 2
    u = torch.arange(num_nodes).repeat_interleave(half)
    off = torch.randint(1, band + 1, (u.numel(),), generator=gen)
    sign = torch.randint(0, 2, (u.numel(),), generator=gen) * 2 - 1
    v = (u + sign * off).clamp_(0, num_nodes - 1)
    keep = u != v
    u, v = u[keep], v[keep]
    ei = torch.stack([torch.cat([u, v]), torch.cat([v, u])])
    x = torch.randn(num_nodes, hidden, generator=gen) if with_x else torch.empty(num_nodes, 0)
    return GraphBatch(x=x, edge_index=ei, graph_ptr=torch.tensor([0, num_nodes]), num_graphs=1)


#: name -> (builder kwargs, hidden d, rank r, adapters per model forward)
WORKLOADS: Dict[str, dict] = {
    "esol":   dict(kind="molecule", mol="esol", graphs=32,   d=300, r=4,  adapters=5),
    "hiv":    dict(kind="molecule", mol="hiv",  graphs=512,  d=300, r=8,  adapters=5),
    "hiv4096": dict(kind="molecule", mol="hiv", graphs=4096, d=300, r=8,  adapters=5),
    "cora":   dict(kind="planetoid", n=2708,  pairs=5278,  d=32,  r=16, adapters=2),
    "pubmed": dict(kind="planetoid", n=19717, pairs=44324, d=32,  r=16, adapters=2),
    "arxiv":  dict(kind="arxiv", n=170_000, pairs=600_000, d=128, r=32, adapters=4),
}


def make_workload(name: str, seed: int = 0, graphs: Optional[int] = None) -> Tuple[GraphBatch, dict]:
    cfg = dict(WORKLOADS[name])
    if graphs is not None and cfg["kind"] == "molecule":
        cfg["graphs"] = graphs
    if cfg["kind"] == "molecule":
        b = molecule_batch(cfg["mol"], cfg["graphs"], cfg["d"], seed)
    elif cfg["kind"] == "planetoid":
        b = planetoid_like(cfg["n"], cfg["pairs"], cfg["d"], seed)
    else:
        b = arxiv_like(cfg["n"], cfg["pairs"], cfg["d"], seed)
    return b, cfg


def shard_graphs(batch: GraphBatch, rank: int, world: int) -> GraphBatch:
    """Contiguous slice of whole graphs for ``rank`` (molecule
    batches are block-diagonal, so a by-graph partition needs no halo)."""
    b = batch.num_graphs
    lo, hi = (b * rank) // world, (b * (rank + 1)) // world
    n0, n1 = int(batch.graph_ptr[lo]), int(batch.graph_ptr[hi])
    src, dst = batch.edge_index[0], batch.edge_index[1]
    keep = (dst >= n0) & (dst < n1)
    ei = batch.edge_index[:, keep] - n0
    return GraphBatch(x=batch.x[..., n0:n1, :].contiguous(), edge_index=ei.contiguous(),
                      graph_ptr=batch.graph_ptr[lo:hi + 1] - n0, num_graphs=hi - lo)
