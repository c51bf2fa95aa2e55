"""Parity of the HIP path (through the C ABI) against the CPU oracle — `-m gpu`.

Tolerance: rtol = atol = 1e-5 against the fp64 oracle (north_star "within 1e-5
fp32"; SURVEY.md §0.4 explains why the relative term is needed for
unnormalised sums). Integer outputs of graph prep are compared exactly.
"""
from __future__ import annotations

import os

import numpy as np
import pytest
import torch

import gconv_adapter_amd as ga
from oracle import dense_ref
from oracle import pyg_restated as oracle

import tile_ref
from helpers import (assert_close, build_module, golden_case, grad_atol, load_golden, run_module)

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------
# K0 graph prep: exact integer parity with a numpy CSR build
# ---------------------------------------------------------------------------
def _numpy_csr(ei: np.ndarray, n: int, normalize: bool, transpose: bool):
    src, dst = ei[0], ei[1]
    if normalize:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    row, col = (src, dst) if transpose else (dst, src)
    order = np.lexsort((col, row))
    row, col = row[order], col[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, row + 1, 1)
    return np.cumsum(rowptr), col


def _random_graph(n, e, seed, symmetric=False, loops=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(n, (e,), generator=g)
    dst = torch.randint(n, (e,), generator=g)
    if not loops:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    if symmetric:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    return torch.stack([src, dst])


@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("n,e,symmetric", [(1, 0, False), (7, 0, False), (5, 9, False),
                                           (300, 2000, True), (5000, 40000, False)])
def test_graph_prep_matches_numpy(dev, n, e, symmetric, normalize):
    ei = _random_graph(n, e, seed=n + e, symmetric=symmetric)
    plan = ga.GraphPlan(ei.to(dev), n, normalize)
    for transpose, csr in ((False, plan.fwd), (True, plan.bwd)):
        rp, col = _numpy_csr(ei.numpy(), n, normalize, transpose)
        assert csr.nnz == rp[-1]
        np.testing.assert_array_equal(csr.rowptr.cpu().numpy(), rp)
        np.testing.assert_array_equal(csr.col.cpu().numpy()[:csr.nnz], col)
    if normalize:
        rp, _ = _numpy_csr(ei.numpy(), n, True, False)
        deg = np.diff(rp) + 1
        np.testing.assert_allclose(plan.dis.cpu().numpy()[:n], deg.astype(np.float64) ** -0.5, rtol=2e-7)
    if symmetric:
        assert plan.symmetric and plan.bwd is plan.fwd


def test_graph_prep_long_rows_and_segments(dev):
    n = 2000
    hub_src = torch.arange(1, 1500)
    ei = torch.stack([torch.cat([hub_src, torch.tensor([5, 6, 7])]),
                      torch.cat([torch.zeros(1499, dtype=torch.long), torch.tensor([9, 9, 9])])])
    plan = ga.GraphPlan(ei.to(dev), n, True)
    assert plan.fwd.n_long == 1 and plan.fwd.max_degree == 1499
    assert plan.fwd.n_seg == -(-1499 // ga._lib.SEG_EDGES)
    assert plan.fwd.long_rows.cpu()[0].item() == 0
    assert not plan.symmetric and plan.bwd.n_long == 0
    assert plan.n_tiles == 0          # the hub row's tile would need a 1499-node halo: not tileable


def _check_tiles(plan, n, transposed=False, order_by=None):
    csr = plan.bwd if transposed else plan.fwd
    T = ga._lib.tile_nodes(n)
    dis = plan.dis.cpu().numpy() if plan.normalize else None
    return tile_ref.check_plan(plan, csr.rowptr.cpu().numpy(), csr.col.cpu().numpy()[:csr.nnz], n, T,
                               ga._lib.tile_slack(T, n), transposed, dis, order_by)


def test_graph_build_tiles(dev):
    """Tiles of a molecule batch against the NumPy restatement of the rule (tests/tile_ref.py), integer
    for integer: cuts near the multiples of the nominal size, on a molecule boundary when one lies
    within a quarter tile below; tiles cut THROUGH a molecule carry their two-hop halo; records sorted
    by rows (halo included), largest first, ties in node order."""
    batch = ga.synth.molecule_batch("hiv", 96, seed=4)
    n = batch.num_nodes
    plan = ga.GraphPlan(batch.edge_index.to(dev), n, True)
    T = ga._lib.tile_nodes(n)
    assert plan.symmetric and plan.n_tiles == -(-n // T) and T == 16
    want = _check_tiles(plan, n)
    tp = plan.tile_ptr.cpu().numpy()
    inner = np.diff(tp[:-1])                         # (the last tile takes what is left)
    assert tp[0] == 0 and tp[-1] == n and (inner > T - T // 4).all() and (inner < T + T // 4).all()
    assert plan.tile_nodes_max == max(r["n0"] for r in want)
    assert plan.struct.tile_edges_max == max(r["ne_own"] + r["ne_halo"] for r in want)
    # a cut is on a molecule boundary (when one lies within the slack: about a third of them at this
    # size) or goes through a molecule, and then the tiles on both sides carry a halo
    gptr = set(batch.graph_ptr.tolist())
    clean = sum(int(q) in gptr for q in tp[1:-1])
    assert 0 < clean < len(tp) - 2
    for t, r in enumerate(want):
        through = (int(tp[t]) not in gptr) or (int(tp[t + 1]) not in gptr)
        assert (r["n0"] > r["count"]) == through, (t, r)      # (molecules are connected)
    assert any(r["n0"] > r["count"] for r in want)     # this batch has molecules larger than a tile
    # every edge of a tile's own rows ends inside the tile or in its 1-hop halo (local ids < n1)
    lcol = plan.tile_lcol().cpu().numpy()
    for r in want:
        own = lcol[r["e0"]:r["e0"] + r["ne_own"]]
        assert ((own >= 0) & (own < r["n1"])).all()
    # a connected random graph: halos outgrow their slots -> not tileable, the general path runs
    big = ga.synth.planetoid_like(3000, 9000, seed=2)
    assert ga.GraphPlan(big.edge_index.to(dev), 3000, True).n_tiles == 0
    # a banded graph (not block-diagonal at all) IS tileable now: every tile carries a halo
    n2 = 3000
    a = torch.arange(n2 - 2)
    chain = torch.stack([torch.cat([a, a]), torch.cat([a + 1, a + 2])])
    ei2 = torch.cat([chain, chain.flip(0)], dim=1)
    plan2 = ga.GraphPlan(ei2.to(dev), n2, True)
    assert plan2.n_tiles == -(-n2 // 16)
    want2 = _check_tiles(plan2, n2)
    assert all(r["n0"] - r["count"] in (4, 8) for r in want2)   # 2 + 2 nodes beyond each open end


def test_graph_build_tiles_of_a_directed_graph(dev):
    """Directed edge list: S and S^T are tiled on the same cuts, each with its own halo (in-neighbours
    for the forward operator, out-neighbours for the transposed one)."""
    n = 900
    a = torch.arange(n - 3)
    ei = torch.stack([torch.cat([a, a + 3]), torch.cat([a + 1, a])])      # i -> i + 1 and i + 3 -> i
    plan = ga.GraphPlan(ei.to(dev), n, True)
    assert not plan.symmetric and plan.n_tiles > 0
    fwd = _check_tiles(plan, n, transposed=False)
    _check_tiles(plan, n, transposed=True, order_by=fwd)
    assert not torch.equal(plan.tile_halo(False)[:, :8], plan.tile_halo(True)[:, :8])


def test_prefetched_plan_equals_direct_build(dev):
    """The plan built ahead of time by the prefetch worker (native thread with the C++
    extension, Python thread without) is the plan a direct build gives: bit-exact index
    arrays, same counters; and the adapter consumes it."""
    batch = ga.synth.molecule_batch("hiv", 64, seed=11)
    n = batch.num_nodes
    ei = batch.edge_index.to(dev)
    direct = ga.GraphPlan(ei, n, True)
    ga.PLAN_CACHE.clear()
    ga.prefetch_plan(ei, n, True)
    plan = ga.get_plan(ei, n, True)
    assert ga.PLAN_CACHE.hits >= 1 and plan is not direct
    for name in ("rowptr", "col"):
        assert torch.equal(getattr(plan.fwd, name), getattr(direct.fwd, name))
    assert torch.equal(plan.tile_ptr, direct.tile_ptr) and torch.equal(plan.tile_order, direct.tile_order)
    assert torch.equal(plan.dis, direct.dis)
    assert (plan.n_tiles, plan.tile_nodes_max, plan.symmetric) == (direct.n_tiles, direct.tile_nodes_max,
                                                                  direct.symmetric)
    # a bad edge list surfaces as RuntimeError when the plan is taken, as in a direct build
    bad = torch.tensor([[0, 1, 70], [1, 0, 2]], device=dev)
    ga.prefetch_plan(bad, 8, True)
    with pytest.raises(RuntimeError):
        ga.get_plan(bad, 8, True)
    ga.PLAN_CACHE.clear()


def test_graph_prep_rejects_bad_index(dev):
    ei = torch.tensor([[0, 1, 7], [1, 0, 2]])
    with pytest.raises(RuntimeError, match="outside"):
        ga.GraphPlan(ei.to(dev), 5, True)
    with pytest.raises(RuntimeError, match="outside"):
        ga.GraphPlan(torch.tensor([[0, -1], [1, 0]]).to(dev), 5, False)


# ---------------------------------------------------------------------------
# K2 SpMM vs oracle propagate
# ---------------------------------------------------------------------------
def _oracle_spmm(h64, ei, n, normalize, transpose=False):
    if transpose:
        S = dense_ref.propagation_matrix(ei.numpy(), n, normalize)
        return torch.from_numpy(S.T @ h64.numpy())
    if normalize:
        e2, w = oracle.gcn_norm(ei, n, torch.float64)
        return oracle.propagate(h64, e2, w, n)
    return oracle.propagate(h64, ei, None, n)


@pytest.mark.parametrize("width", [4, 8, 16, 32, 64])
@pytest.mark.parametrize("normalize", [True, False])
def test_spmm_matches_oracle(dev, width, normalize):
    n = 700
    ei = _random_graph(n, 4000, seed=width, symmetric=False)
    # a hub row (> LONG_ROW edges, several segments) and an isolated tail
    hub = torch.stack([torch.randint(n - 50, (400,), generator=torch.Generator().manual_seed(1)),
                       torch.full((400,), 3)])
    ei = torch.cat([ei[:, (ei[0] < n - 50) & (ei[1] < n - 50)], hub], dim=1)
    h = torch.randn(n, width, generator=torch.Generator().manual_seed(2))
    plan = ga.GraphPlan(ei.to(dev), n, normalize)
    assert plan.fwd.n_long >= 1
    for transpose in (False, True):
        y = ga.spmm(plan, h.to(dev), transpose=transpose)
        want = _oracle_spmm(h.double(), ei, n, normalize, transpose)
        assert_close(y.cpu().double(), want, f"spmm F={width} norm={normalize} T={transpose}")


def test_spmm_bias_act_and_prescale_flags(dev):
    n, f = 257, 16
    ei = _random_graph(n, 1500, seed=5)
    h = torch.randn(n, f, generator=torch.Generator().manual_seed(6))
    b = torch.randn(f, generator=torch.Generator().manual_seed(7))
    plan = ga.GraphPlan(ei.to(dev), n, True)
    want = _oracle_spmm(h.double(), ei, n, True) + b.double()
    for act, fn in (("relu", torch.relu), ("silu", torch.nn.functional.silu), ("none", lambda t: t)):
        y = ga.spmm(plan, h.to(dev), bias=b.to(dev), act=act)
        assert_close(y.cpu().double(), fn(want), f"spmm act={act}")
    # prescaled input / output chaining: dis * S(h) computed from dis*h
    dis = plan.dis[:n]
    y = ga.spmm(plan, (h.to(dev) * dis[:, None]), flags=ga._lib.SPMM_IN_PRESCALED | ga._lib.SPMM_OUT_PRESCALE)
    want2 = _oracle_spmm(h.double(), ei, n, True) * dis.cpu().double()[:, None]
    assert_close(y.cpu().double(), want2, "spmm prescaled chain")


def test_spmm_is_bitwise_deterministic(dev):
    b, _ = ga.synth.make_workload("pubmed")
    n = b.num_nodes
    plan = ga.GraphPlan(b.edge_index.to(dev), n, True)
    h = torch.randn(n, 16, device=dev)
    y0 = ga.spmm(plan, h)
    for _ in range(3):
        assert torch.equal(ga.spmm(plan, h), y0)


# ---------------------------------------------------------------------------
# K1 / K3 projections
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,r", [(1, 8, 1), (17, 300, 4), (100, 300, 8), (333, 128, 32),
                                   (64, 32, 16), (50, 64, 64), (1000, 300, 5), (40, 1024, 16)])
def test_gemm_down_and_up(dev, n, d, r):
    g = torch.Generator().manual_seed(n * d + r)
    x = torch.randn(n, d, generator=g)
    wd = torch.randn(r, d, generator=g) * 0.1
    wu = torch.randn(d, r, generator=g) * 0.1
    bu = torch.randn(d, generator=g)
    scale = torch.rand(n, generator=g) + 0.5
    s = torch.tensor([0.7])
    h0 = ga.gemm_down(x.to(dev), wd.to(dev), scale.to(dev))
    fp = h0.shape[1]
    want = (x.double() @ wd.double().T) * scale.double()[:, None]
    assert_close(h0[:, :r].cpu().double(), want, "gemm_down")
    assert torch.count_nonzero(h0[:, r:]) == 0, "padding columns must be zero"
    p = torch.zeros(n, fp)
    p[:, :r] = torch.randn(n, r, generator=g)
    out = ga.gemm_up_fused(p.to(dev), wu.to(dev), bu.to(dev), s.to(dev), x.to(dev), skip=True)
    want = 0.7 * (p[:, :r].double() @ wu.double().T + bu.double() + x.double())
    assert_close(out.cpu().double(), want, "gemm_up_fused")
    out = ga.gemm_up_fused(p.to(dev), wu.to(dev), None, None, None, skip=False)
    assert_close(out.cpu().double(), p[:, :r].double() @ wu.double().T, "gemm_up no-bias no-skip")


# ---------------------------------------------------------------------------
# Whole adapter vs golden fixtures (module surface, autograd backward)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g1_path5_norm.npz", "g1_path5_raw.npz", "g2_molecules.npz",
                                  "g3_transductive.npz"])
def test_adapter_golden(dev, name):
    case = golden_case(load_golden(name))
    d = case["x"].shape[-1]
    r = case["params"]["conv_down.lin.weight"].shape[0]
    mod = build_module(ga, d, r, case["flags"], case["params"], dev)
    got = run_module(mod, case["x"], case["edge_index"], case["grad_out"], dev)
    for k, want in case["expected"].items():
        assert_close(got[k], want, f"{name}:{k}", atol=1e-5 if k in ("out", "x") else grad_atol(want))


def test_adapter_golden_flag_grid(dev):
    rec = load_golden("g4_flags.npz")
    for i in range(rec["x"].shape[0]):
        case = golden_case(rec, i)
        mod = build_module(ga, 16, 4, case["flags"], case["params"], dev)
        got = run_module(mod, case["x"], case["edge_index"], case["grad_out"], dev)
        for k, want in case["expected"].items():
            atol = 1e-5 if k in ("out", "x") else grad_atol(want)
            if case["flags"].normalization == "batch_norm" and k == "conv_up.bias":
                # BatchNorm removes the column mean, so this gradient is exactly 0 in exact
                # arithmetic; what is left is fp32 cancellation noise of torch's own BatchNorm
                # backward (its output can be O(1e3) when the pre-norm std is small). Bound it
                # by the noise the fp32 CPU oracle shows on the same case.
                _, g32 = oracle.adapter_forward_backward(
                    case["x"].float(), case["edge_index"],
                    {kk: v.float() for kk, v in case["params"].items()}, case["flags"],
                    case["grad_out"].float())
                atol = 1e-5 + 20.0 * float((g32[k].double() - want).abs().max())
            assert_close(got[k], want, f"g4[{i}] {case['flags']}:{k}", atol=atol)


# ---------------------------------------------------------------------------
# Whole adapter vs the oracle on seeded workloads of the benchmark shapes
# ---------------------------------------------------------------------------
def _check_workload(dev, batch, d, r, flags, seed=0, need_dx=True, rtol=1e-5, atol=1e-5):
    params = {k: v.float().double() for k, v in oracle.init_params(d, r, flags, seed=seed).items()}
    g = torch.randn(batch.x.shape, generator=torch.Generator().manual_seed(seed + 1))
    want_out, want_grads = oracle.adapter_forward_backward(batch.x.double(), batch.edge_index, params,
                                                           flags, g.double(), need_dx=need_dx)
    mod = build_module(ga, d, r, flags, params, dev)
    got = run_module(mod, batch.x, batch.edge_index, g, dev, need_dx=need_dx)
    assert_close(got["out"], want_out, "out", rtol, atol)
    for k, want in want_grads.items():
        assert_close(got[k], want, f"grad {k}", rtol, atol if k == "x" else grad_atol(want, atol))


@pytest.mark.parametrize("normalize", [True, False])
def test_adapter_esol_batch(dev, normalize):
    batch, cfg = ga.synth.make_workload("esol")
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=normalize)
    _check_workload(dev, batch, cfg["d"], cfg["r"], flags)


@pytest.mark.parametrize("normalize", [True, False])
def test_adapter_hiv_shard(dev, normalize):
    batch, cfg = ga.synth.make_workload("hiv", graphs=128)
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=normalize)
    _check_workload(dev, batch, cfg["d"], cfg["r"], flags)


@pytest.mark.parametrize("d,r,act,normalize", [
    (8, 1, "relu", True), (16, 3, "silu", True), (64, 4, "relu", False), (100, 8, "none", True),
    (128, 16, "relu", True), (200, 5, "silu", False), (256, 8, "relu", True), (300, 16, "silu", True),
    (300, 32, "relu", True), (304, 2, "relu", False), (320, 8, "relu", True), (384, 16, "none", True),
    (512, 4, "relu", True), (512, 8, "silu", False)])
def test_adapter_tile_path_width_rank_sweep(dev, d, r, act, normalize):
    """The fused tile kernels at every compiled row length / rank class (ring sizes, co-owner
    slot maps and the range-checked column tails differ per (d, Fp)), vs the fp64 oracle."""
    batch = ga.synth.molecule_batch("hiv", 40, hidden=d, seed=d + r)
    flags = oracle.AdapterFlags(non_linearity=act, learnable_scalar=True, normalize=normalize)
    _check_workload(dev, batch, d, r, flags, seed=r)


@pytest.mark.skipif(bool(os.environ.get("GCONV_NO_TILE_PATH")), reason="asserts that the tile path is taken")
@pytest.mark.parametrize("chords_per_node", [16, 2])
def test_adapter_tile_path_large_components(dev, chords_per_node):
    """Components far larger than a tile (400-node rings with chords) next to single-node and two-node
    graphs: with a chord per 16 nodes every tile is cut through its component and carries a halo (ring
    neighbours two hops out plus chord ends anywhere in the ring and THEIR neighbours); with a chord
    per 2 nodes the halos outgrow their slots and the whole batch takes the general path."""
    gen = torch.Generator().manual_seed(5)
    sizes = [400, 1, 2, 390, 17, 1, 350]
    src, dst, off = [], [], 0
    for n in sizes:
        if n > 1:
            a = torch.arange(n)
            ring = torch.stack([a, (a + 1) % n]) if n > 2 else torch.tensor([[0], [1]])
            chords = torch.randint(n, (2, n // chords_per_node), generator=gen)
            chords = chords[:, chords[0] != chords[1]]
            e = torch.cat([ring, chords], dim=1)
            e = torch.cat([e, e.flip(0)], dim=1) + off
            src.append(e[0]); dst.append(e[1])
        off += n
    ei = torch.stack([torch.cat(src), torch.cat(dst)])
    x = torch.randn(off, 300, generator=gen)
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    batch = ga.synth.GraphBatch(x=x, edge_index=ei, graph_ptr=ptr, num_graphs=len(sizes))
    plan = ga.GraphPlan(ei.to(dev), off, True)
    lib = ga._lib.load()
    if chords_per_node == 16:
        # the fused tile path takes the batch (forward workspace: the 16-byte placeholder)
        assert plan.n_tiles > 0 and 16 < plan.tile_nodes_max <= 20 + 64
        assert plan.sizes(lib, 300, 8)[1] == 16
        assert (plan.tiles()[:, 3] > plan.tiles()[:, 1]).sum() >= 30      # tiles with a halo
    else:
        assert plan.n_tiles == 0 and plan.sizes(lib, 300, 8)[1] > 16
    for normalize in (True, False):
        flags = oracle.AdapterFlags(learnable_scalar=True, normalize=normalize)
        _check_workload(dev, batch, 300, 8, flags)
    if chords_per_node != 16:
        return
    # rank 16 (the reference's YAML default) with a 230-atom molecule in the batch: 87 KB of LDS
    keep = torch.tensor([i for i in range(len(sizes))])
    sizes2 = [230, 12, 1, 25, 31]
    src, dst, off2 = [], [], 0
    for n in sizes2:
        if n > 1:
            a = torch.arange(n)
            e = torch.stack([a, (a + 1) % n])
            e = torch.cat([e, e.flip(0)], dim=1) + off2
            src.append(e[0]); dst.append(e[1])
        off2 += n
    ei2 = torch.stack([torch.cat(src), torch.cat(dst)])
    batch2 = ga.synth.GraphBatch(x=torch.randn(off2, 300, generator=gen), edge_index=ei2,
                                 graph_ptr=torch.tensor([0] + list(np.cumsum(sizes2))), num_graphs=len(sizes2))
    plan2 = ga.GraphPlan(ei2.to(dev), off2, True)
    assert plan2.n_tiles > 0 and plan2.sizes(lib, 300, 16)[1] == 16
    _check_workload(dev, batch2, 300, 16, oracle.AdapterFlags(learnable_scalar=True, normalize=True))


def test_adapter_many_tiles_per_workgroup(dev):
    """3200 molecules = 768 tiles of ~106 nodes on 256 persistent workgroups: every workgroup walks
    3 tiles (boustrophedon passes over the size-sorted tile list, records requested a pass ahead,
    accumulators carried across tiles) — vs the fp64 oracle."""
    batch = ga.synth.molecule_batch("hiv", 3200, hidden=300, seed=21)
    plan = ga.GraphPlan(batch.edge_index.to(dev), batch.num_nodes, True)
    assert plan.n_tiles >= 3 * 256 and (plan.tiles()[:-1, 3] >= plan.tiles()[1:, 3]).all()   # size-sorted
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=True)
    _check_workload(dev, batch, 300, 8, flags, seed=5)


@pytest.mark.parametrize("seed", range(24))
def test_adapter_random_configurations(dev, seed):
    """Seeded random sweep over what the reference can be handed: component structure (one blob,
    many small components, isolated nodes), directed / symmetrised edge lists with duplicates
    and self-loops, hidden width, rank, activation, normalisation, skip, scalar — vs the oracle.
    Covers both execution paths (block-diagonal inputs take the fused tile kernels)."""
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    ncomp = [1, 1, 3, 12, 40][seed % 5]
    sizes = [ri(1, max(1, [150, 40, 30, 12, 6][seed % 5])) for _ in range(ncomp)]
    src, dst, off = [], [], 0
    for n in sizes:
        e = ri(0, 4 * n)
        if e and n > 0:
            a = torch.randint(n, (e,), generator=g) + off
            b = torch.randint(n, (e,), generator=g) + off      # self-loops and duplicates welcome
            src.append(a); dst.append(b)
        off += n
    n = off
    ei = torch.stack([torch.cat(src), torch.cat(dst)]) if src else torch.zeros(2, 0, dtype=torch.long)
    if seed % 3 != 0 and ei.shape[1]:
        ei = torch.cat([ei, ei.flip(0)], dim=1)                 # symmetrised (duplicates kept)
    d = 4 * ri(1, 16)
    r = ri(1, min(16, d))
    flags = oracle.AdapterFlags(non_linearity=["relu", "silu", "none"][seed % 3],
                                learnable_scalar=seed % 2 == 0, skip_connection=seed % 4 != 3,
                                normalize=seed % 5 != 4)
    x = torch.randn(n, d, generator=g)
    batch = ga.synth.GraphBatch(x=x, edge_index=ei, graph_ptr=torch.tensor([0, n]), num_graphs=1)
    _check_workload(dev, batch, d, r, flags, seed=seed)


def test_adapter_cora_3d_input(dev):
    batch, cfg = ga.synth.make_workload("cora")
    assert batch.x.dim() == 3
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=True)
    _check_workload(dev, batch, cfg["d"], cfg["r"], flags)


def test_adapter_power_law_graph_rank32(dev):
    # scaled-down S-arxiv: hubs with thousands of edges (long-row path), d=128, r=32
    batch = ga.synth.arxiv_like(num_nodes=20_000, num_pairs=70_000, hidden=128, seed=3)
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=True)
    _check_workload(dev, batch, 128, 32, flags)


def test_adapter_asymmetric_graph_uses_transposed_csr(dev):
    n = 500
    ei = _random_graph(n, 3000, seed=9, symmetric=False)
    x = torch.randn(n, 64, generator=torch.Generator().manual_seed(10))
    batch = ga.synth.GraphBatch(x=x, edge_index=ei, graph_ptr=torch.tensor([0, n]), num_graphs=1)
    for normalize in (True, False):
        for act in ("relu", "silu", "none"):
            flags = oracle.AdapterFlags(non_linearity=act, learnable_scalar=True, normalize=normalize)
            _check_workload(dev, batch, 64, 12, flags)


def test_adapter_without_input_grad_and_without_skip(dev):
    batch, cfg = ga.synth.make_workload("esol")
    flags = oracle.AdapterFlags(learnable_scalar=False, skip_connection=False, normalize=True)
    _check_workload(dev, batch, cfg["d"], cfg["r"], flags, need_dx=False)


def test_adapter_edge_cases(dev):
    # empty edge list, single node, all-isolated nodes
    for n, ei in ((1, torch.zeros(2, 0, dtype=torch.long)), (6, torch.zeros(2, 0, dtype=torch.long)),
                  (3, torch.tensor([[0, 0, 1], [0, 0, 1]]))):
        x = torch.randn(n, 8, generator=torch.Generator().manual_seed(n))
        batch = ga.synth.GraphBatch(x=x, edge_index=ei, graph_ptr=torch.tensor([0, n]), num_graphs=1)
        for normalize in (True, False):
            flags = oracle.AdapterFlags(learnable_scalar=True, normalize=normalize)
            _check_workload(dev, batch, 8, 2, flags)


def test_adapter_zero_nodes_and_device_mismatch(dev):
    mod = ga.GConvAdapter(8, 2, learnable_scalar=True).to(dev)
    x = torch.zeros(0, 8, device=dev, requires_grad=True)
    out = mod(x, torch.zeros(2, 0, dtype=torch.long, device=dev))
    assert out.shape == (0, 8)
    out.sum().backward()
    assert all(torch.count_nonzero(p.grad) == 0 for p in mod.parameters())
    with pytest.raises(RuntimeError, match="edge_index is on"):
        mod(torch.randn(2, 8, device=dev), torch.tensor([[0, 1], [1, 0]]))


def test_adapter_near_identity_at_reference_init(dev):
    # reference init (gconv_adapter.py:65,73-78): W ~ N(0, 1e-5), b = 0, s = 1 -> out ~= x
    batch, cfg = ga.synth.make_workload("esol")
    mod = ga.GConvAdapter(cfg["d"], cfg["r"], learnable_scalar=True).to(dev)
    x = batch.x.to(dev)
    out = mod(x, batch.edge_index.to(dev))
    assert (out - x).abs().max().item() < 1e-6


def test_adapter_rejects_cpu_and_wrong_dtype(dev):
    mod = ga.GConvAdapter(8, 2).to(dev)
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(RuntimeError, match="no CPU path"):
        mod(torch.randn(2, 8), ei)
    with pytest.raises(RuntimeError, match="fp32"):
        mod(torch.randn(2, 8, device=dev).double(), ei.to(dev))
    with pytest.raises(RuntimeError):
        ga.GConvAdapter(10, 2).to(dev)(torch.randn(2, 10, device=dev), ei.to(dev))  # d % 4 != 0


@pytest.mark.parametrize("workload", ["esol", "cora"])
def test_split_backward_with_batched_finalize_is_bit_identical(dev, workload):
    """C ABI: gconv_adapter_bwd_partial_f32 x 3 + ONE gconv_adapter_bwd_finalize_f32 == three
    gconv_adapter_bwd_f32 calls, bit for bit (fused tile path: esol; general path: cora)."""
    import ctypes as C
    lib = ga._lib.load()
    batch, cfg = ga.synth.make_workload(workload)
    d, r = cfg["d"], cfg["r"]
    x = batch.x.reshape(-1, d).to(dev); ei = batch.edge_index.to(dev)
    n = x.shape[0]
    plan = ga.get_plan(ei, n, True)
    fp, wsf, wsb = plan.sizes(lib, d, r)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(3)
    results = {"whole": [], "split": []}
    pend = (ga._lib.PendingFinalize * 3)()
    keep = []
    for mode in ("whole", "split"):
        for k in range(3):
            g = torch.Generator().manual_seed(10 + k)
            wd = (torch.randn(r, d, generator=g) * .1).to(dev); bd = (torch.randn(r, generator=g) * .1).to(dev)
            wu = (torch.randn(d, r, generator=g) * .1).to(dev); bu = (torch.randn(d, generator=g) * .1).to(dev)
            sc = torch.full((1,), 0.7 + 0.1 * k, device=dev)
            go = torch.randn(n, d, generator=g).to(dev)
            P = ga._lib.AdapterParams(wd.data_ptr(), bd.data_ptr(), wu.data_ptr(), bu.data_ptr(), sc.data_ptr(),
                                      d, r, 1, 1)
            out = torch.empty_like(x); saved = torch.empty(2, n, fp, device=dev)
            wF = torch.empty(wsf, dtype=torch.uint8, device=dev); wB = torch.empty(wsb, dtype=torch.uint8, device=dev)
            ga._lib.check(lib.gconv_adapter_fwd_f32(plan.ref(), C.byref(P), x.data_ptr(), out.data_ptr(),
                                                    saved[0].data_ptr(), None, saved[1].data_ptr(),
                                                    wF.data_ptr(), wsf, st), "fwd")
            gx = torch.empty_like(x)
            gp = [torch.empty_like(t) for t in (wd, bd, wu, bu, sc)]
            args = (plan.ref(), C.byref(P), x.data_ptr(), go.data_ptr(), saved[0].data_ptr(), None,
                    saved[1].data_ptr(), gx.data_ptr(), gp[0].data_ptr(), gp[1].data_ptr(), gp[2].data_ptr(),
                    gp[3].data_ptr(), gp[4].data_ptr(), wB.data_ptr(), wsb)
            if mode == "whole":
                ga._lib.check(lib.gconv_adapter_bwd_f32(*args, st), "bwd")
            else:
                ga._lib.check(lib.gconv_adapter_bwd_partial_f32(*args, C.byref(pend[k]), st), "bwd_partial")
            keep.append((wd, bd, wu, bu, sc, go, saved, wB, P))
            results[mode].append([gx] + gp)
    ga._lib.check(lib.gconv_adapter_bwd_finalize_f32(pend, 3, st), "bwd_finalize")
    torch.cuda.synchronize()
    for a, b in zip(results["whole"], results["split"]):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)
    assert lib.gconv_adapter_bwd_finalize_f32(None, 0, st) == 0
    assert lib.gconv_adapter_bwd_finalize_f32(None, 2, st) == -1     # GCONV_E_INVALID_ARG


def test_direct_param_grad_deposit_equals_autograd_accumulation(dev):
    """C++ binding, two ways of storing the parameter gradients — written into p.grad by the
    node itself (default) or returned to AccumulateGrad nodes: bit-identical results; a second
    backward accumulates; inputs without grad still train the parameters; a parameter with a
    tensor hook takes the standard route."""
    if ga._torch_ext.get() is None:
        pytest.skip("pure-Python binding selected")
    batch, cfg = ga.synth.make_workload("esol")
    d, r = cfg["d"], cfg["r"]
    x = batch.x.to(dev); ei = batch.edge_index.to(dev)
    g = torch.randn(x.shape, generator=torch.Generator().manual_seed(4)).to(dev)
    mods = [ga.GConvAdapter(d, r, learnable_scalar=True, non_linearity="silu").to(dev) for _ in range(2)]
    with torch.no_grad():
        for p0, p1 in zip(mods[0].parameters(), mods[1].parameters()):
            p0.normal_(0, 0.1); p1.copy_(p0)
    mods[1].direct_param_grads = False
    res = []
    for m in mods:
        xg = x.clone().requires_grad_(True)
        out = m(xg, ei)
        out.backward(g, retain_graph=True)
        first = [p.grad.clone() for p in m.parameters()]
        out.backward(g)                       # accumulates into the existing .grad
        res.append((out.detach(), xg.grad, first, [p.grad for p in m.parameters()]))
    assert type(res[0][0].grad_fn) is type(None)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)
    for first, twice in zip(res[0][2], res[0][3]):
        assert torch.equal(twice, first + first)
    # x without grad: the anchor keeps the node alive
    m = mods[0]
    m.zero_grad(set_to_none=True)
    m(x, ei).backward(g)
    for p, want in zip(m.parameters(), res[0][2]):   # (another kernel variant: no dX)
        assert p.grad is not None and torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6)
    # a tensor hook on a parameter -> standard autograd (the hook runs)
    m2 = ga.GConvAdapter(d, r, learnable_scalar=True).to(dev)
    seen = []
    m2.conv_up.bias.register_hook(lambda gr: seen.append(gr.shape))
    m2(x, ei).backward(g)
    assert seen == [m2.conv_up.bias.shape] and m2.conv_down.lin.weight.grad is not None


def test_adapter_step_is_hip_graph_capturable(dev):
    """Static graphs (the transductive setting): with the plan built beforehand, forward + backward
    of a chain of adapters make no allocation outside torch's allocator, no synchronisation and
    no host read-back, so the whole step can be captured in a HIP graph and replayed — same bits."""
    if ga._torch_ext.get() is None:
        pytest.skip("capture is exercised with the C++ binding")
    batch, cfg = ga.synth.make_workload("cora")
    d, r = cfg["d"], cfg["r"]
    x = batch.x.to(dev); ei = batch.edge_index.to(dev)
    gout = torch.randn(x.shape, generator=torch.Generator().manual_seed(2)).to(dev)
    mods = [ga.GConvAdapter(d, r, learnable_scalar=True).to(dev) for _ in range(2)]
    with torch.no_grad():
        for m in mods:
            for p in m.parameters():
                p.normal_(0, 0.1)
    params = [p for m in mods for p in m.parameters()]
    ga.get_plan(ei, x.shape[-2], True)

    def step():
        for p in params:
            p.grad = None
        h = x
        for m in mods:
            h = m(h, ei)
        h.backward(gout)
        return h

    out_ref = step().detach().clone()
    ref = [p.grad.clone() for p in params]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    for p in params:
        p.grad = None
    with torch.cuda.graph(graph):
        h = x
        for m in mods:
            h = m(h, ei)
        h.backward(gout)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(h, out_ref)
    for p, want in zip(params, ref):
        assert torch.equal(p.grad, want)


def test_plan_cache_shared_across_adapters(dev):
    batch, cfg = ga.synth.make_workload("esol")
    ga.PLAN_CACHE.clear()
    h0, m0 = ga.PLAN_CACHE.hits, ga.PLAN_CACHE.misses
    ei = batch.edge_index.to(dev)
    x = batch.x.to(dev)
    mods = [ga.GConvAdapter(cfg["d"], cfg["r"]).to(dev) for _ in range(5)]
    for m in mods:
        x = m(x, ei)
    assert ga.PLAN_CACHE.misses - m0 == 1 and ga.PLAN_CACHE.hits - h0 == 4
    ei[0, 0] = ei[0, 0]  # in-place write bumps _version -> new plan
    mods[0](x, ei)
    assert ga.PLAN_CACHE.misses - m0 == 2


# ---------------------------------------------------------------------------
# Full benchmark size: size-independent properties (oracle too slow / fp64 too big)
# ---------------------------------------------------------------------------
def test_device_collation_equals_cpu_collation(dev):
    """MoleculeStore.collate (two launches on the device) == concatenating the chosen graphs on
    the CPU the way PyG's Batch.from_data_list does: bit-exact features, edge ids, offsets and
    node -> graph vector; repeated and empty selections; the adapter runs on the result."""
    whole = ga.synth.molecule_batch("hiv", 300, hidden=64, seed=13)
    store = ga.MoleculeStore.from_batch(whole, dev)
    ptr = whole.graph_ptr
    src, dst = whole.edge_index
    gid = torch.bucketize(dst, ptr[1:], right=True)
    g = torch.Generator().manual_seed(1)
    for ids in (torch.randperm(300, generator=g)[:77].tolist(), [5, 5, 299, 0], list(range(300)), []):
        xs, es, gp, bv = [], [], [0], []
        for p, k in enumerate(ids):
            n0, n1 = int(ptr[k]), int(ptr[k + 1])
            m = gid == k                                    # the graph's edges, in dataset order
            xs.append(whole.x[n0:n1])
            es.append(torch.stack([src[m], dst[m]]) - n0 + gp[-1])
            bv += [p] * (n1 - n0)
            gp.append(gp[-1] + n1 - n0)
        want_x = torch.cat(xs) if xs else torch.empty(0, 64)
        want_e = torch.cat(es, dim=1) if es else torch.empty(2, 0, dtype=torch.long)
        batch, bvec = store.collate(ids, with_batch_vector=True)
        assert torch.equal(batch.x.cpu(), want_x) and torch.equal(batch.edge_index.cpu(), want_e)
        assert batch.graph_ptr.cpu().tolist() == gp and bvec.cpu().tolist() == bv
    # the batch's graph plan, assembled from the dataset's plan (no sort, no sync) == the plan
    # gconv_graph_build makes of the collated edge list
    ids = torch.randperm(300, generator=g)[:64].tolist()
    for normalize in (True, False):
        ga.PLAN_CACHE.clear()
        batch = store.collate(ids, normalize=normalize)
        fast = ga.get_plan(batch.edge_index, batch.num_nodes, normalize)
        assert ga.PLAN_CACHE.misses == 0 or fast.arena is not None
        direct = ga.GraphPlan(batch.edge_index, batch.num_nodes, normalize)
        assert fast is not direct and fast.fwd.nnz == direct.fwd.nnz and fast.symmetric
        assert torch.equal(fast.fwd.rowptr, direct.fwd.rowptr) and torch.equal(fast.fwd.col, direct.fwd.col)
        # its tiles end on graph boundaries (no halo: nothing to read back), a direct build's may not
        ft = fast.tiles().cpu().numpy()
        assert fast.n_tiles == direct.n_tiles and (ft[:, 3] == ft[:, 1]).all() and ft[:, 1].sum() == batch.num_nodes
        assert set(fast.tile_ptr.cpu().tolist()) <= set(batch.graph_ptr.cpu().tolist())
        assert fast.tile_nodes_max >= ft[:, 3].max()
        if normalize:
            assert torch.equal(fast.dis, direct.dis)
        mod = ga.GConvAdapter(64, 4, learnable_scalar=True, normalize=normalize).to(dev)
        with torch.no_grad():
            for prm in mod.parameters():
                prm.normal_(0, 0.1)
        xg = batch.x.clone().requires_grad_(True)
        out = mod(xg, batch.edge_index)                       # runs on the assembled plan
        out.sum().backward()
        ei2 = batch.edge_index.clone()                        # another tensor: plan built directly
        x2 = batch.x.clone().requires_grad_(True)
        grads = [prm.grad.clone() for prm in mod.parameters()]
        mod.zero_grad(set_to_none=True)
        ref = mod(x2, ei2)
        ref.sum().backward()
        # a row's result does not depend on the tiling (bit-equal); the parameter gradients are sums
        # over the tiles, and the two plans cut the batch differently
        assert torch.equal(out, ref) and torch.equal(xg.grad, x2.grad)
        for ga_, gb_ in zip(grads, [prm.grad for prm in mod.parameters()]):
            assert torch.allclose(ga_, gb_, rtol=1e-5, atol=1e-5 * float(gb_.abs().max()))
    with pytest.raises(RuntimeError):
        store.collate([300])


def test_fused_adam_matches_torch_adam(dev):
    """FusedAdam (one launch per group over flat buffers) vs torch.optim.Adam on the CPU in
    fp64 and on the GPU in fp32: two parameter groups with their own lr / weight decay (the
    reference's head group, finetune_inductive_learning.py:375-379), a frozen parameter, six
    steps; state_dict round trip into a torch Adam."""
    torch.manual_seed(0)
    shapes = [(8, 300), (8,), (300, 8), (300,), (1,), (300, 2)]
    base = [torch.randn(s) * 0.1 for s in shapes]
    grads = [[torch.randn(s) for s in shapes] for _ in range(6)]

    def make(device, dtype):
        ps = [torch.nn.Parameter(b.to(device=device, dtype=dtype).clone()) for b in base]
        frozen = torch.nn.Parameter(torch.ones(3, device=device, dtype=dtype), requires_grad=False)
        groups = [{"params": ps[:5] + [frozen]}, {"params": ps[5:], "lr": 3e-3, "weight_decay": 0.0}]
        return ps, groups

    ps_ref, groups_ref = make("cpu", torch.float64)
    opt_ref = torch.optim.Adam(groups_ref, lr=1e-3, weight_decay=1e-2)
    ps_t, groups_t = make(dev, torch.float32)
    opt_t = torch.optim.Adam(groups_t, lr=1e-3, weight_decay=1e-2)
    ps_f, groups_f = make(dev, torch.float32)
    opt_f = ga.FusedAdam(groups_f, lr=1e-3, weight_decay=1e-2)
    for step in range(6):
        for ps, opt in ((ps_ref, opt_ref), (ps_t, opt_t), (ps_f, opt_f)):
            for p, g in zip(ps, grads[step]):
                p.grad = g.to(device=p.device, dtype=p.dtype)
            opt.step()
    for pr, pt, pf in zip(ps_ref, ps_t, ps_f):
        assert torch.allclose(pf.detach().cpu().double(), pr.detach(), rtol=1e-5, atol=1e-7)
        assert torch.allclose(pf.detach(), pt.detach(), rtol=1e-6, atol=1e-7)   # 1e-7 = 3e-5 of one step
    # the parameters are views of one flat buffer per group, still the same Parameter objects
    assert ps_f[0].data_ptr() + 4 * ps_f[0].numel() == ps_f[1].data_ptr()
    # state_dict is torch-Adam compatible: continue in a torch Adam, one more identical step
    import copy
    sd = copy.deepcopy(opt_f.state_dict())   # (state_dict() hands out references, as torch's does)
    ps_c, groups_c = make(dev, torch.float32)
    with torch.no_grad():
        for pc, pf in zip(ps_c, ps_f):
            pc.copy_(pf)
    opt_c = torch.optim.Adam(groups_c, lr=1e-3, weight_decay=1e-2)
    opt_c.load_state_dict(sd)
    g7 = [torch.randn(s) for s in shapes]
    for ps, opt in ((ps_c, opt_c), (ps_f, opt_f)):
        for p, g in zip(ps, g7):
            p.grad = g.to(dev)
        opt.step()
    for pc, pf in zip(ps_c, ps_f):
        assert torch.allclose(pc.detach(), pf.detach(), rtol=1e-6, atol=1e-7), \
            f"max abs diff {(pc - pf).abs().max().item():.3e} on {tuple(pc.shape)}"
    # and the other way round: a torch state dict loaded into a FusedAdam
    ps_g, groups_g = make(dev, torch.float32)
    with torch.no_grad():
        for pg, pc in zip(ps_g, ps_c):
            pg.copy_(pc)
    opt_g = ga.FusedAdam(groups_g, lr=1e-3, weight_decay=1e-2)
    opt_g.load_state_dict(copy.deepcopy(opt_c.state_dict()))
    g8 = [torch.randn(s) for s in shapes]
    for ps, opt in ((ps_c, opt_c), (ps_g, opt_g)):
        for p, g in zip(ps, g8):
            p.grad = g.to(dev)
        opt.step()
    for pc, pg in zip(ps_c, ps_g):
        assert torch.allclose(pc.detach(), pg.detach(), rtol=1e-6, atol=1e-7)


def test_long_row_completion_under_load(dev):
    """Long rows are cut into segments summed by different waves (different XCDs, private L2s);
    the wave that arrives last at the row's counter adds the partial sums. Repeated at
    S-arxiv scale with another stream hammering memory: every launch must match the fp64
    reference (a stale partial would show as an O(1) error), and the counters must be back at
    zero for the next launch."""
    batch, cfg = ga.synth.make_workload("arxiv")
    n, r = batch.num_nodes, cfg["r"]
    ei = batch.edge_index.to(dev)
    plan = ga.get_plan(ei, n, True)
    assert plan.fwd.n_long > 100 and plan.fwd.n_seg > plan.fwd.n_long
    dis = plan.dis[:n].double()
    side = torch.cuda.Stream()
    hog = torch.randn(32 << 20, device=dev)
    g = torch.Generator().manual_seed(0)
    for it in range(12):
        h = torch.randn(n, r, generator=g).to(dev)
        with torch.cuda.stream(side):
            for _ in range(4):
                hog.mul_(1.0001)
        y = ga.spmm(plan, h)
        hd = h.double() * dis[:, None]
        ref = dis[:, None] * (torch.zeros(n, r, device=dev, dtype=torch.float64).index_add_(0, ei[1], hd[ei[0]]) + hd)
        assert (y.double() - ref).abs().max().item() < 1e-4, f"launch {it}"
    torch.cuda.synchronize()
    sync = plan.arena_view(plan.fwd.struct.long_sync, plan.fwd.n_long, torch.int32)
    assert int(sync.abs().sum()) == 0


def test_arxiv_scale_properties(dev):
    batch, cfg = ga.synth.make_workload("arxiv")
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    ei = batch.edge_index.to(dev)
    plan = ga.get_plan(ei, n, True)
    assert plan.symmetric and plan.fwd.n_long > 0
    g = torch.Generator().manual_seed(0)
    h1 = torch.randn(n, r, generator=g).to(dev)
    h2 = torch.randn(n, r, generator=g).to(dev)
    # linearity of the SpMM
    lhs = ga.spmm(plan, 2.0 * h1 - 3.0 * h2)
    rhs = 2.0 * ga.spmm(plan, h1) - 3.0 * ga.spmm(plan, h2)
    assert torch.allclose(lhs, rhs, rtol=1e-5, atol=1e-5)
    # S is symmetric for this edge list: <S a, b> == <a, S b>
    a, b = h1.double(), h2.double()
    assert abs((ga.spmm(plan, h1).double() * b).sum() - (a * ga.spmm(plan, h2).double()).sum()).item() \
        < 1e-6 * n
    # constant vector: S 1 row sums equal dis_i * sum_j dis_j over the looped neighbourhood
    ones = torch.ones(n, 4, device=dev)
    rowsum = ga.spmm(plan, ones)[:, 0]
    dis = plan.dis[:n]
    dd = dis.double()
    ref = dd * (torch.zeros(n, device=dev, dtype=torch.float64).index_add_(0, ei[1], dd[ei[0]]) + dd)
    assert torch.allclose(rowsum.double(), ref, rtol=1e-5, atol=1e-6)
    # module level: bitwise run-to-run determinism of forward and backward
    mod = ga.GConvAdapter(d, r, learnable_scalar=True).to(dev)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, 0.1)
    x = batch.x.to(dev).requires_grad_(True)
    gout = torch.randn(n, d, generator=g).to(dev)
    outs = []
    for _ in range(2):
        mod.zero_grad(set_to_none=True)
        x.grad = None
        out = mod(x, ei)
        out.backward(gout)
        outs.append([out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in mod.parameters()])
    for t0, t1 in zip(*outs):
        assert torch.equal(t0, t1)
    # fp32 oracle in the reassociated order on the CPU (same math, 1e-5 bar at full size)
    params = {k: v.detach().cpu() for k, v in mod.named_parameters()}
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=True)
    want = oracle.adapter_forward_reassociated(batch.x.double(), batch.edge_index,
                                               {k: v.double() for k, v in params.items()}, flags)
    assert_close(outs[0][0].cpu().double(), want, "arxiv-scale out")


# ---------------------------------------------------------------------------
# layer_norm tail: LayerNorm + scalar in one pass each way
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("shape,affine,with_scalar", [
    ((1, 8), True, True), ((5, 300), True, True), ((3000, 300), True, True), ((1, 700, 32), True, True),
    ((257, 1024), True, False), ((64, 128), False, True), ((33, 516), False, False)])
def test_layer_norm_scale_matches_torch_fp64(dev, shape, affine, with_scalar):
    from gconv_adapter_amd.functional import LayerNormScaleFunction
    d = shape[-1]
    g = torch.Generator().manual_seed(d + shape[0])
    z = (3.0 * torch.randn(shape, generator=g) + 1.5)
    gamma = (1.0 + 0.3 * torch.randn(d, generator=g)) if affine else None
    beta = (0.3 * torch.randn(d, generator=g)) if affine else None
    s = torch.tensor([1.3]) if with_scalar else None
    go = torch.randn(shape, generator=g)
    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (z, gamma, beta, s)]
    want = torch.nn.functional.layer_norm(leaves[0], (d,), leaves[1], leaves[2], 1e-5)
    if s is not None:
        want = want * leaves[3]
    want.backward(go.double())
    dl = [t.to(dev).requires_grad_(True) if t is not None else None for t in (z, gamma, beta, s)]
    got = LayerNormScaleFunction.apply(dl[0], dl[1], dl[2], dl[3], 1e-5)
    got.backward(go.to(dev))
    assert got.shape == z.shape
    assert_close(got.detach().cpu().double(), want.detach(), "out")
    assert_close(dl[0].grad.cpu().double(), leaves[0].grad, "grad z", 1e-5, 1e-5)
    for name, a, b in (("gamma", dl[1], leaves[1]), ("beta", dl[2], leaves[2]), ("scalar", dl[3], leaves[3])):
        if a is not None:
            assert_close(a.grad.cpu().double(), b.grad, f"grad {name}", 1e-5, grad_atol(b.grad))
    again = LayerNormScaleFunction.apply(dl[0], dl[1], dl[2], dl[3], 1e-5)
    assert torch.equal(again, got)


def test_adapter_layer_norm_uses_the_fused_tail(dev):
    """The module path with normalization='layer_norm' (ablation 4) against the oracle, 2-D and
    [1, N, d] inputs, with and without the learnable scalar."""
    batch, cfg = ga.synth.make_workload("esol")
    for scalar in (True, False):
        flags = oracle.AdapterFlags(normalization="layer_norm", learnable_scalar=scalar, normalize=True)
        _check_workload(dev, batch, cfg["d"], cfg["r"], flags)
    cora, ccfg = ga.synth.make_workload("cora")
    flags = oracle.AdapterFlags(normalization="layer_norm", learnable_scalar=True, normalize=True)
    _check_workload(dev, cora, ccfg["d"], ccfg["r"], flags)


# ---------------------------------------------------------------------------
# BatchNorm1d over node features (the backbone's norm after every GIN layer)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,relu,shift", [(2, 8, False, 0.0), (431, 300, False, 0.0), (13104, 300, True, 0.0),
                                            (1000, 1024, False, 0.0), (5000, 64, True, 50.0)])
def test_batch_norm_matches_torch_fp64(dev, n, d, relu, shift):
    g = torch.Generator().manual_seed(n + d)
    x = 2.0 * torch.randn(n, d, generator=g) + shift + torch.randn(d, generator=g)
    go = torch.randn(n, d, generator=g)
    ref = torch.nn.BatchNorm1d(d).double()
    mine = ga.BatchNorm1d(d, fused_relu=relu).to(dev)
    with torch.no_grad():
        ref.weight.copy_(1.0 + 0.3 * torch.randn(d, generator=g))
        ref.bias.copy_(0.3 * torch.randn(d, generator=g))
    mine.load_state_dict({k: (v.float() if v.dtype.is_floating_point else v) for k, v in ref.state_dict().items()})
    for step in range(3):   # running statistics accumulate over steps
        xr = (x + 0.1 * step).double().requires_grad_(True)
        yr = ref(xr)
        if relu:
            yr = torch.relu(yr)
        ref.zero_grad()
        yr.backward(go.double())
        xm = (x + 0.1 * step).to(dev).requires_grad_(True)
        ym = mine(xm)
        mine.zero_grad()
        ym.backward(go.to(dev))
    # a large common offset (shift) costs relative accuracy in x - mean itself: (|mean| / std) * 2^-24
    tol = 1e-5 * max(1.0, shift)
    assert_close(ym.detach().cpu().double(), yr.detach(), "y", 1e-5, tol)
    assert_close(xm.grad.cpu().double(), xr.grad, "grad x", 1e-5, tol)
    assert_close(mine.weight.grad.cpu().double(), ref.weight.grad, "grad gamma", 1e-5, grad_atol(ref.weight.grad) * max(1.0, shift))
    assert_close(mine.bias.grad.cpu().double(), ref.bias.grad, "grad beta", 1e-5, grad_atol(ref.bias.grad))
    assert_close(mine.running_mean.cpu().double(), ref.running_mean, "running_mean", 1e-5, 1e-5 * max(1.0, shift))
    assert_close(mine.running_var.cpu().double(), ref.running_var, "running_var", 1e-5, tol)
    assert int(mine.num_batches_tracked) == 3
    assert set(mine.state_dict()) == set(ref.state_dict())
    ref.eval(); mine.eval()
    with torch.no_grad():
        ye = ref(x.double())
        ye = torch.relu(ye) if relu else ye
        assert_close(mine(x.to(dev)).cpu().double(), ye, "eval y", 1e-5, tol)
    xe = x.to(dev).requires_grad_(True)          # eval mode with autograd: torch's formula
    mine(xe).sum().backward()
    assert torch.isfinite(xe.grad).all()
    again = ga.BatchNorm1d(d, fused_relu=relu).to(dev).train()
    again.load_state_dict(mine.state_dict())
    a, b = again(x.to(dev)), ga.norm.BatchNormFunction.apply(x.to(dev), again.weight, again.bias, None, None, 0.0,
                                                             again.eps, relu)
    assert torch.equal(a, b)                      # bitwise repeatable


@pytest.mark.parametrize("relu", [False, True])
def test_batch_norm_cpp_node_equals_the_python_node(dev, relu):
    """BatchNorm1d's autograd node exists twice — csrc/torch_binding.cpp BatchNormFn (one host call; what the
    module uses when the binding is built) and norm.BatchNormFunction (ctypes) — around the same C ABI calls:
    outputs, running statistics and all three gradients agree bit for bit."""
    if ga._torch_ext.get() is None:
        pytest.skip("the C++ binding is not built (or GCONV_PY_BINDING is set)")
    g = torch.Generator().manual_seed(5)
    n, d = 777, 300
    x0 = (torch.randn(n, d, generator=g) * 2 + 1).to(dev)
    go = torch.randn(n, d, generator=g).to(dev)
    res = []
    for node in ("cpp", "python"):
        m = ga.BatchNorm1d(d, fused_relu=relu).to(dev).train()
        with torch.no_grad():
            m.weight.copy_(torch.linspace(0.5, 1.5, d)); m.bias.copy_(torch.linspace(-1, 1, d))
        x = x0.clone().requires_grad_(True)
        if node == "cpp":
            y = m(x)
        else:
            y = ga.norm.BatchNormFunction.apply(x, m.weight, m.bias, m.running_mean, m.running_var, m.momentum,
                                                m.eps, relu)
        y.backward(go)
        res.append((y.detach(), x.grad, m.weight.grad, m.bias.grad, m.running_mean.clone(), m.running_var.clone()))
        if node == "cpp":   # num_batches_tracked is counted by the kernel (no launch of its own)
            m(x0); m.eval(); m(x0)
            assert int(m.num_batches_tracked) == 2 and m.num_batches_tracked.dtype == torch.int64
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_batch_norm_other_inputs_take_torchs_path(dev):
    m = ga.BatchNorm1d(6).to(dev)                 # d % 4 != 0 -> nn.BatchNorm1d's own kernels
    x = torch.randn(10, 6, device=dev)
    assert torch.allclose(m(x), torch.nn.functional.batch_norm(x, None, None, m.weight, m.bias, True), atol=1e-6)
    with pytest.raises(ValueError, match="more than 1 value"):
        ga.BatchNorm1d(8).to(dev)(torch.randn(1, 8, device=dev))


# ---------------------------------------------------------------------------
# larger batches: several tiles of equal nominal size per persistent workgroup
# ---------------------------------------------------------------------------
def test_tiles_of_whole_batches(dev):
    """2700 HIV-sized molecules (~69 k nodes, 3 tiles per workgroup): nominal tile size n / (256 k) (`_lib.tile_nodes`), same
    rule, same results — direct build vs the restatement, plan assembled from a resident dataset,
    adapter vs the oracle."""
    batch = ga.synth.molecule_batch("hiv", 2700, hidden=64, seed=12)
    n = batch.num_nodes
    T = ga._lib.tile_nodes(n)
    assert n > 256 * 128 and 64 < T <= 128 and T == -(-n // (256 * -(-(-(-n // 256)) // 128)))
    ei = batch.edge_index.to(dev)
    plan = ga.GraphPlan(ei, n, True)
    assert plan.symmetric and plan.n_tiles == -(-n // T)
    _check_tiles(plan, n)
    store = ga.MoleculeStore.from_batch(batch, dev)           # the same batch, assembled from the dataset plan
    got = store.collate(np.arange(batch.num_graphs), normalize=True)
    plan2 = ga.get_plan(got.edge_index, n, True)
    assert plan2.n_tiles == plan.n_tiles and torch.equal(plan2.fwd.col, plan.fwd.col)
    ga.PLAN_CACHE.clear()
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=True)
    _check_workload(dev, batch, 64, 8, flags)


@pytest.mark.parametrize("kind,d,r,normalize", [("cora", 32, 16, True), ("cora", 32, 16, False),
                                                ("hub", 128, 32, True), ("hub", 300, 8, True),
                                                ("hub", 64, 4, False), ("directed", 128, 32, True)])
def test_fused_gather_launches_equal_separate_launches(dev, kind, d, r, normalize, monkeypatch):
    """General path: K2' + K3 and B3 + B4 as ONE launch each (fused_gather.hip; rows gathered inside
    the projection kernels, long rows by the SpMM's segment blocks in front) must give the bits of
    the separate launches — out, every gradient (the saved P feeds the backward) — for hub graphs with
    long rows, directed graphs, both normalisation modes, ragged last node blocks."""
    if kind == "cora":
        batch, _ = ga.synth.make_workload("cora")
        ei, n = batch.edge_index, batch.num_nodes
    elif kind == "hub":
        n = 3001
        g = torch.Generator().manual_seed(5)
        u = torch.randint(n, (9000,), generator=g)
        v = (n * torch.rand(9000, generator=g) ** 3).long().clamp_(max=n - 1)
        hub = torch.stack([torch.randint(n, (700,), generator=g), torch.zeros(700, dtype=torch.long)])
        pairs = torch.cat([torch.stack([u, v]), hub], dim=1)
        ei = torch.cat([pairs, pairs.flip(0)], dim=1)
    else:
        n = 2500
        ei = _random_graph(n, 12000, 9, symmetric=False)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(1)).to(dev)
    gr = torch.randn(n, d, generator=torch.Generator().manual_seed(2)).to(dev)
    eid = ei.to(dev)
    mod = ga.GConvAdapter(d, r, learnable_scalar=True, normalize=normalize, non_linearity="silu").to(dev)
    pg = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn(p.shape, generator=pg) * 0.1)
    res = []
    monkeypatch.setenv("GCONV_FUSED_GATHER_MAX_NODES", "1000000")   # also for graphs with long rows
    for fused in ("0", "1"):
        monkeypatch.setenv("GCONV_NO_FUSED_GATHER", "0" if fused == "1" else "1")
        mod.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        out = mod(xg, eid)
        out.backward(gr)
        res.append([out.detach(), xg.grad] + [p.grad for p in mod.parameters()])
    names = ["out", "dx"] + [k for k, _ in mod.named_parameters()]
    for name, a, b in zip(names, *res):
        assert torch.equal(a, b), f"{name}: {int((a != b).sum())} elements differ, max {float((a - b).abs().max()):.3e}"
    if kind == "hub":
        assert ga.get_plan(eid, n, normalize).fwd.n_long > 0


def test_batch_norm_tail_with_folded_scalar(dev):
    """`normalization='batch_norm'` + `learnable_scalar`: the scalar rides in the affine pair of the
    BatchNorm kernels (s (gamma xhat + beta) = (s gamma) xhat + (s beta)) — same out and gradients
    (incl. d scalar) as BatchNorm followed by `* scalar` in fp64, training and evaluation mode."""
    torch.manual_seed(3)
    n, d = 431, 300
    x = torch.randn(n, d) * 2 + 0.5
    g = torch.randn(n, d)
    m = ga.BatchNorm1d(d).to(dev)
    ref = torch.nn.BatchNorm1d(d).double()
    with torch.no_grad():
        m.weight.normal_(1, 0.2); m.bias.normal_(0, 0.2)
        ref.weight.copy_(m.weight.cpu()); ref.bias.copy_(m.bias.cpu())
    s = torch.tensor([0.6], device=dev, requires_grad=True)
    s64 = torch.tensor([0.6], dtype=torch.float64, requires_grad=True)
    for training in (True, False):
        m.train(training); ref.train(training)
        for t in (m.weight, m.bias, s, ref.weight, ref.bias, s64):
            t.grad = None
        xd = x.to(dev).requires_grad_(True)
        x64 = x.double().requires_grad_(True)
        out = m(xd, s)
        out.backward(g.to(dev))
        want = ref(x64) * s64
        want.backward(g.double())
        assert_close(out.detach().cpu().double(), want.detach(), "out")
        assert_close(xd.grad.cpu().double(), x64.grad, "dx")
        for got, w, name in ((m.weight.grad, ref.weight.grad, "dgamma"), (m.bias.grad, ref.bias.grad, "dbeta"),
                             (s.grad, s64.grad, "dscalar")):
            assert_close(got.cpu().double(), w, name, atol=grad_atol(w))
    assert torch.allclose(m.running_mean.cpu().double(), ref.running_mean, atol=1e-6)


@pytest.mark.parametrize("d,r,act", [(1024, 16, "relu"), (64, 64, "silu"), (128, 33, "none"), (512, 64, "relu")])
def test_adapter_general_path_extreme_shapes(dev, d, r, act):
    """Largest hidden size, largest rank, a rank that pads to 64 — on a connected random graph (no
    component tiles, so the general path: K1, K2, K2'(+K3 fused), B1..B5)."""
    n = 700
    ei = _random_graph(n, 5000, 21, symmetric=True)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(3))
    batch = ga.synth.GraphBatch(x=x, edge_index=ei, graph_ptr=torch.tensor([0, n]), num_graphs=1)
    flags = oracle.AdapterFlags(non_linearity=act, learnable_scalar=True, normalize=True)
    _check_workload(dev, batch, d, r, flags)
    assert ga.get_plan(ei.to(dev), n, True).n_tiles == 0
    if d == 512:   # beyond the LDS of a CU (hidden x padded rank > 40960 floats): a RuntimeError, not a fault
        with pytest.raises(RuntimeError, match="unsupported shape"):
            ga.GConvAdapter(1024, 64).to(dev)(torch.randn(n, 1024, device=dev), ei.to(dev))


def test_adapter_hub_with_ten_thousand_edges(dev):
    """One row with > 10^4 edges (S-arxiv's node 0 has 10,875): ~85 long-row segments summed by
    different waves and completed in-kernel, through the whole adapter forward and backward."""
    n, d, r = 12000, 64, 8
    g = torch.Generator().manual_seed(17)
    spokes = torch.stack([torch.arange(1, 10501), torch.zeros(10500, dtype=torch.long)])
    rnd = torch.stack([torch.randint(n, (20000,), generator=g), torch.randint(n, (20000,), generator=g)])
    pairs = torch.cat([spokes, rnd], dim=1)
    ei = torch.cat([pairs, pairs.flip(0)], dim=1)
    x = torch.randn(n, d, generator=g)
    batch = ga.synth.GraphBatch(x=x, edge_index=ei, graph_ptr=torch.tensor([0, n]), num_graphs=1)
    for normalize in (True, False):
        flags = oracle.AdapterFlags(learnable_scalar=True, normalize=normalize)
        _check_workload(dev, batch, d, r, flags)
    plan = ga.get_plan(ei.to(dev), n, True)
    assert plan.fwd.max_degree > 10000 and plan.fwd.n_seg >= 80


def test_adapter_batched_3d_and_non_contiguous_inputs(dev):
    """[B, N, d] with B > 1 (node_dim = -2: every batch entry is propagated over the same graph,
    SURVEY 8c item 6), a non-contiguous x (a transposed view) and a non-contiguous edge_index."""
    batch, cfg = ga.synth.make_workload("cora")
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=True)
    params = {k: v.float().double() for k, v in oracle.init_params(d, r, flags, seed=4).items()}
    mod = build_module(ga, d, r, flags, params, dev)
    gen = torch.Generator().manual_seed(8)
    x3 = torch.randn(3, n, d, generator=gen)
    g3 = torch.randn(3, n, d, generator=gen)
    want_out, want = oracle.adapter_forward_backward(x3.double(), batch.edge_index, params, flags, g3.double())
    xd = x3.to(dev).requires_grad_(True)
    mod.zero_grad(set_to_none=True)
    out = mod(xd, batch.edge_index.to(dev))
    out.backward(g3.to(dev))
    assert_close(out.detach().cpu().double(), want_out, "out [3, N, d]")
    assert_close(xd.grad.cpu().double(), want["x"], "dx [3, N, d]")
    for k, p in mod.named_parameters():
        assert_close(p.grad.cpu().double(), want[k], f"grad {k}", atol=grad_atol(want[k]))
    # B > 1 runs as ONE call over B disjoint copies of the graph: per entry the numbers of the 2-D call (1e-5: the
    # tiles of the repeated graph are cut elsewhere), one plan for all calls with the same list, and an endpoint
    # outside [0, N) still fails instead of reaching into the next copy
    eid = batch.edge_index.to(dev)
    with torch.no_grad():
        o3 = mod(xd.detach(), eid)
        misses = ga.graph.PLAN_CACHE.misses
        o3b = mod(xd.detach(), eid)
        assert ga.graph.PLAN_CACHE.misses == misses and torch.equal(o3, o3b)
        for b in range(3):
            assert_close(o3[b].cpu().double(), mod(xd.detach()[b], eid).cpu().double(), f"entry {b} vs 2-D call")
        bad = eid.clone()
        bad[1, 5] = n
        with pytest.raises(RuntimeError, match="outside"):
            mod(xd.detach(), bad)
        assert mod(xd.detach().transpose(0, 1).contiguous().transpose(0, 1), eid).shape == (3, n, d)   # strided [B, N, d]
    # non-contiguous x / edge_index: same numbers as their contiguous copies
    xt = torch.randn(d, n, generator=gen).to(dev).t()                 # [n, d] view with strides (1, n)
    ei_nc = torch.stack([batch.edge_index[0], batch.edge_index[1]], dim=1).to(dev).t()   # [2, E] view of [E, 2]
    assert not xt.is_contiguous() and not ei_nc.is_contiguous()
    with torch.no_grad():
        a = mod(xt, ei_nc)
        b = mod(xt.contiguous(), batch.edge_index.to(dev))
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="int64"):
        mod(xt.contiguous(), batch.edge_index.to(dev).int())


def test_adapter_large_weight_table_on_many_rows(dev):
    """hidden 512 x rank 64: K1's weight-fragment table takes 128 KB of LDS (one workgroup per CU),
    on enough rows (20000) that the persistent grid is capped by the CU count — the number of
    d(scalar) partials B1 writes must be the number the finalize launch sums."""
    n, d, r = 20000, 512, 64
    ei = _random_graph(n, 60000, 31, symmetric=True)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(5))
    batch = ga.synth.GraphBatch(x=x, edge_index=ei, graph_ptr=torch.tensor([0, n]), num_graphs=1)
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=True)
    _check_workload(dev, batch, d, r, flags)


def _plan_arrays(plan):
    """Everything a plan holds, long rows keyed by row (their order in the lists is not specified)."""
    def csr(v):
        s = v.struct
        out = {"nnz": v.nnz, "rowptr": v.rowptr.cpu(), "col": v.col.cpu(), "n_long": v.n_long, "n_seg": v.n_seg,
               # (the single-workgroup builder of graphs up to 4096 nodes does not count the near edges)
               "near_edges": v.near_edges if plan.num_nodes > 4096 else None}
        if v.n_long:
            rows = plan.arena_view(s.long_rows, v.n_long, torch.int32).cpu().tolist()
            base = plan.arena_view(s.long_seg_base, v.n_long, torch.int32).cpu().tolist()
            seg = plan.arena_view(s.seg_long, v.n_seg, torch.int32).cpu().tolist()
            rp = out["rowptr"].tolist()
            per_row = {}
            for lr, (r, b) in enumerate(zip(rows, base)):
                nseg = -(-(rp[r + 1] - rp[r]) // 128)
                assert seg[b:b + nseg] == [lr] * nseg       # the row's segments point back at it
                per_row[r] = nseg
            assert sorted(b for b in base) == sorted(set(base)) and sum(per_row.values()) == v.n_seg
            out["long"] = per_row
        return out
    res = {"fwd": csr(plan.fwd), "symmetric": plan.symmetric, "n_tiles": plan.n_tiles,
           "max_degree": plan.struct.max_degree}
    if not plan.symmetric:
        res["bwd"] = csr(plan.bwd)
    if plan.normalize:
        res["dis"] = plan.dis.cpu()
    if plan.n_tiles:
        res["tile_ptr"] = plan.tile_ptr.cpu()
        res["tile_max"] = (plan.struct.tile_nodes_max, plan.struct.tile_edges_max)
        for name, tr in (("fwd", False),) + ((("bwd", True),) if not plan.symmetric else ()):
            recs, lcol, halo = plan.tiles(tr).cpu(), plan.tile_lcol(tr).cpu(), plan.tile_halo(tr).cpu()
            res["tiles_" + name] = recs
            res["lcol_" + name] = lcol
            used = {}
            for first, count, n1, n0, slot, e0, ne, nhe in recs.tolist():   # the slots' defined prefixes
                h = halo[slot]
                used[slot] = (h[:n0 - count].tolist(), h[64:64 + n1 - count + 1].tolist() if n1 > count else [],
                              h[132:132 + min(nhe, 256)].tolist(),
                              h[388:388 + n0 - count].tolist() if plan.normalize else [])
            res["halo_" + name] = used
    return res


def _same(a, b, path=""):
    assert type(a) is type(b), path
    if isinstance(a, dict):
        assert a.keys() == b.keys(), (path, a.keys(), b.keys())
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, torch.Tensor):
        assert torch.equal(a, b), path
    else:
        assert a == b, (path, a, b)


@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("kind", ["esol", "hiv32", "random", "random_2k", "directed", "hub", "empty", "single",
                                  "loops_only", "wide_edge", "max_size"])
def test_single_workgroup_graph_build_equals_multi_launch_build(dev, kind, normalize, monkeypatch):
    """Graphs of up to 4096 nodes / 4096 edges are built by ONE workgroup in ONE launch
    (small_graph_build_kernel: keys, bitonic sort, CSR, dis, long rows, exact symmetry test, S^T for
    directed lists, component tiles, size-sorted tile list — all in LDS): every array and counter
    must equal the multi-launch builder's (long-row lists up to their order)."""
    g = torch.Generator().manual_seed(41)
    if kind == "esol":
        b, _ = ga.synth.make_workload("esol"); ei, n = b.edge_index, b.num_nodes
    elif kind == "hiv32":
        b, _ = ga.synth.make_workload("hiv", graphs=32); ei, n = b.edge_index, b.num_nodes
    elif kind == "random":
        n = 777; ei = _random_graph(n, 1500, 42, symmetric=True)          # duplicates and self-loops included
    elif kind == "random_2k":   # 256-thread variant at its largest key count
        n = 300; ei = _random_graph(n, 2040, 45, symmetric=False)
    elif kind == "directed":
        n = 500; ei = _random_graph(n, 4000, 43, symmetric=False)
    elif kind == "hub":
        n = 1500
        spokes = torch.stack([torch.randint(1, n, (900,), generator=g), torch.zeros(900, dtype=torch.long)])
        rnd = torch.stack([torch.randint(n, (1000,), generator=g), torch.randint(n, (1000,), generator=g)])
        pairs = torch.cat([spokes, rnd], dim=1)
        ei = torch.cat([pairs, pairs.flip(0)], dim=1)
    elif kind == "empty":
        n = 40; ei = torch.zeros(2, 0, dtype=torch.long)
    elif kind == "single":
        n = 1; ei = torch.tensor([[0, 0], [0, 0]])
    elif kind == "loops_only":
        n = 9; ei = torch.arange(9).repeat(2, 1)
    elif kind == "wide_edge":      # an edge across the whole graph: its ends become halo nodes of each other's tiles
        n = 2000; ei = torch.tensor([[0, 1999, 5, 6], [1999, 0, 6, 5]])
    else:                          # the size limits themselves
        n = 4096; ei = _random_graph(n, 2048, 44, symmetric=True)
        assert ei.shape[1] == 4096
    eid = ei.to(dev)
    plans = []
    for off in ("1", "0"):
        monkeypatch.setenv("GCONV_NO_SMALL_BUILD", off)
        plans.append(_plan_arrays(ga.GraphPlan(eid, n, normalize)))
    _same(plans[0], plans[1])
    if kind == "hub":
        assert plans[1]["fwd"]["n_long"] >= 1
    if kind == "directed":
        assert not plans[1]["symmetric"]
    # a bad endpoint is reported, never a fault, by both builders
    if ei.shape[1]:
        bad = ei.clone(); bad[0, 0] = n
        for off in ("1", "0"):
            monkeypatch.setenv("GCONV_NO_SMALL_BUILD", off)
            with pytest.raises(RuntimeError, match="outside"):
                ga.GraphPlan(bad.to(dev), n, normalize)


@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("kind", ["hiv512", "hiv2000", "random_sym", "dups_and_loops", "directed", "hub", "bad"])
def test_counting_graph_build_equals_radix_build(dev, kind, normalize, monkeypatch):
    """Graphs of 4097 .. 262144 nodes are built by COUNTING (five launches: count, scan, fill, row
    sort, symmetry test + tiles) instead of the radix-sort pipeline; a hub row or a directed list makes
    the host fall back to it. Every array and counter must equal the radix builder's."""
    g = torch.Generator().manual_seed(7)
    if kind == "hiv512":
        b, _ = ga.synth.make_workload("hiv"); ei, n = b.edge_index, b.num_nodes
    elif kind == "hiv2000":
        b, _ = ga.synth.make_workload("hiv", graphs=2000); ei, n = b.edge_index, b.num_nodes
    elif kind == "random_sym":
        n = 9000; ei = _random_graph(n, 20000, 51, symmetric=True, loops=False)
    elif kind == "dups_and_loops":
        n = 5000; ei = _random_graph(n, 6000, 52, symmetric=True)
        ei = torch.cat([ei, ei[:, :500], torch.arange(100).repeat(2, 1)], dim=1)     # duplicates, self-loops
    elif kind == "directed":
        n = 6000; ei = _random_graph(n, 15000, 53, symmetric=False)
    elif kind == "hub":
        n = 7000
        spokes = torch.stack([torch.randint(1, n, (900,), generator=g), torch.zeros(900, dtype=torch.long)])
        rnd = torch.stack([torch.randint(n, (5000,), generator=g), torch.randint(n, (5000,), generator=g)])
        pairs = torch.cat([spokes, rnd], dim=1)
        ei = torch.cat([pairs, pairs.flip(0)], dim=1)
    else:
        n = 5000; ei = _random_graph(n, 9000, 54, symmetric=True); ei[1, 17] = n + 3
    eid = ei.to(dev)
    if kind == "bad":
        for off in ("1", "0"):
            monkeypatch.setenv("GCONV_NO_COUNTING_BUILD", off)
            with pytest.raises(RuntimeError, match="outside"):
                ga.GraphPlan(eid, n, normalize)
        return
    plans = []
    for off in ("1", "0"):
        monkeypatch.setenv("GCONV_NO_COUNTING_BUILD", off)
        plans.append(_plan_arrays(ga.GraphPlan(eid, n, normalize)))
    _same(plans[0], plans[1])
    rp, col = _numpy_csr(ei.numpy(), n, normalize, False)
    np.testing.assert_array_equal(plans[1]["fwd"]["rowptr"].numpy(), rp)
    np.testing.assert_array_equal(plans[1]["fwd"]["col"].numpy()[:rp[-1]], col)
    if kind == "hub":
        assert plans[1]["fwd"]["n_long"] >= 1
    if kind == "directed":
        assert not plans[1]["symmetric"]


@pytest.mark.parametrize("hub_degree", [16, 17])
def test_counting_graph_build_row_capacity(dev, hub_degree, monkeypatch):
    """The counting builder keeps a row's edges in 16 slots: a row of exactly 16 is built by it, one of
    17 makes the host repeat the build on the radix pipeline. Same plan either way."""
    n = 5000
    ring = torch.stack([torch.arange(n), (torch.arange(n) + 1) % n])
    spokes = torch.stack([torch.arange(100, 100 + hub_degree - 2) * 7, torch.full((hub_degree - 2,), 3)])
    pairs = torch.cat([ring, spokes], dim=1)
    ei = torch.cat([pairs, pairs.flip(0)], dim=1).to(dev)
    plans = []
    for off in ("1", "0"):
        monkeypatch.setenv("GCONV_NO_COUNTING_BUILD", off)
        plans.append(_plan_arrays(ga.GraphPlan(ei, n, True)))
    _same(plans[0], plans[1])
    assert plans[1]["max_degree"] == hub_degree


def test_counting_graph_build_from_two_threads(dev):
    """The one-launch builder keeps its barrier and counter words in library-owned device memory that
    must read all zeros between launches; concurrent builds (the collation thread and the training
    loop) are serialised on a mutex. Two threads, own streams, alternating sizes: every plan equals the
    one built alone."""
    import threading
    batches = [ga.synth.make_workload("hiv", graphs=g)[0] for g in (300, 512)]
    eis = [b.edge_index.to(dev) for b in batches]
    want = [_plan_arrays(ga.GraphPlan(ei, b.num_nodes, True)) for ei, b in zip(eis, batches)]
    errors = []

    def work(k):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for it in range(25):
                    j = (it + k) % 2
                    got = _plan_arrays(ga.GraphPlan(eis[j], batches[j].num_nodes, True))
                    _same(want[j], got)
        except Exception as exc:   # noqa: BLE001 - reported by the main thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]


@pytest.mark.parametrize("mode", ["default", "split", "fused", "radix"])
def test_background_graph_build_equals_foreground_build(dev, mode, monkeypatch):
    """GraphPlan(background=True) (GCONV_BUILD_BACKGROUND: the prefetch worker's builds, next to the step's
    kernels) never waits inside a wide kernel: the sort pipeline (default / `radix`: measured to disturb the
    step least), the counting builder as one NARROW launch per phase (`split`) or its narrow one-launch form
    (`fused`). Same plan every way, also for the graphs that make the counting builder hand over to the
    radix pipeline."""
    if mode != "default":
        monkeypatch.setenv("GCONV_BG_BUILD", mode)
    cases = []
    for graphs in (512, 300, 2000):
        b, _ = ga.synth.make_workload("hiv", graphs=graphs)
        cases.append((b.edge_index, b.num_nodes))
    cases.append((_random_graph(6000, 15000, 53, symmetric=False), 6000))                    # directed: radix after all
    dl = _random_graph(5000, 6000, 52, symmetric=True)
    cases.append((torch.cat([dl, dl[:, :500], torch.arange(100).repeat(2, 1)], dim=1), 5000))  # duplicates, self-loops
    for ei, n in cases:
        eid = ei.to(dev)
        for normalize in (True, False):
            fg = _plan_arrays(ga.GraphPlan(eid, n, normalize))
            for _ in range(2):   # twice: the library-owned counters must be back at zero after a split build
                _same(fg, _plan_arrays(ga.GraphPlan(eid, n, normalize, background=True)))
    bad = cases[0][0].clone(); bad[1, 17] = cases[0][1] + 3
    with pytest.raises(RuntimeError, match="outside"):
        ga.GraphPlan(bad.to(dev), cases[0][1], True, background=True)
    _same(_plan_arrays(ga.GraphPlan(cases[0][0].to(dev), cases[0][1], True)),
          _plan_arrays(ga.GraphPlan(cases[0][0].to(dev), cases[0][1], True, background=True)))   # and fine after the refusal


def test_cooperative_launch_of_the_counting_build(dev, monkeypatch):
    """GCONV_COUNT_COOPERATIVE=1: the one-launch builder goes through hipLaunchCooperativeKernel (the runtime
    checks that the whole grid can be resident, or refuses the launch -> radix pipeline). Same plan."""
    b, _ = ga.synth.make_workload("hiv")
    ei = b.edge_index.to(dev)
    want = _plan_arrays(ga.GraphPlan(ei, b.num_nodes, True))
    monkeypatch.setenv("GCONV_COUNT_COOPERATIVE", "1")
    for _ in range(3):
        _same(want, _plan_arrays(ga.GraphPlan(ei, b.num_nodes, True)))
