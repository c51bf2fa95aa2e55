"""The frozen GIN layer next to the adapter: oracle self-checks on CPU,
parity of the HIP aggregation (through the module and the C ABI) against the oracle on GPU."""
from __future__ import annotations

import numpy as np
import pytest
import torch

import gconv_adapter_amd as ga
from oracle import gin_restated as oracle

from helpers import assert_close, grad_atol


def _bond_attr(num_edges: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randint(0, 4, (num_edges,), generator=g),
                        torch.randint(0, 3, (num_edges,), generator=g)], dim=1)


# --------------------------------------------------------------------------- CPU
def test_oracle_edge_list_form_equals_dense_form():
    batch = ga.synth.molecule_batch("esol", 6, hidden=12, seed=3)
    ea = _bond_attr(batch.edge_index.shape[1], 1)
    p = oracle.init_params(12, seed=2)
    x = batch.x.double()
    a = oracle.gin_aggregate(x, batch.edge_index, ea, p)
    b = oracle.gin_aggregate_dense(x, batch.edge_index, ea, p)
    assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)


def test_oracle_self_loop_semantics():
    """gin_conv.py:50-55: every node gets one extra (i, i) edge of bond type 4, direction 0 —
    an isolated node's aggregate is x_i + T_type[4] + T_dir[0]; an existing loop is kept."""
    p = oracle.init_params(4, seed=0)
    x = torch.randn(3, 4, dtype=torch.float64)
    ei = torch.tensor([[0, 1], [1, 1]])            # 0 -> 1 and an explicit loop on 1; node 2 isolated
    ea = torch.tensor([[2, 1], [0, 2]])
    t, d = p["edge_embedding_type.weight"], p["edge_embedding_direction.weight"]
    agg = oracle.gin_aggregate(x, ei, ea, p)
    assert torch.allclose(agg[2], x[2] + t[4] + d[0])
    assert torch.allclose(agg[0], x[0] + t[4] + d[0])
    assert torch.allclose(agg[1], x[0] + t[2] + d[1] + x[1] + t[0] + d[2] + x[1] + t[4] + d[0])


def test_module_surface_matches_the_reference():
    m = ga.MolecularGINConv(300)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {
        "mlp.0.weight": (600, 300), "mlp.0.bias": (600,), "mlp.2.weight": (300, 600), "mlp.2.bias": (300,),
        "edge_embedding_type.weight": (6, 300), "edge_embedding_direction.weight": (3, 300)}
    with pytest.raises(ValueError):
        ga.MolecularGINConv(300, aggr="mean")
    with pytest.raises(ValueError):
        ga.MolecularGINConv(302)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(2, 300), torch.zeros(2, 0, dtype=torch.long), torch.zeros(0, 2, dtype=torch.long))


def test_module_is_complete_after_construction_and_pickles():
    """Round-4 advisor findings: the fp16-pair cache used to appear only in `_apply` / `load_state_dict`
    (a module built directly on its device never had it), and a lambda hook made the module unpicklable."""
    import copy
    import pickle
    m = ga.MolecularGINConv(32)
    assert len(m.__dict__["_mlp_pair"]) == 4 and len(m.__dict__["_mlp_split"]) == 4
    m2 = pickle.loads(pickle.dumps(m))                      # what torch.save(model) does
    assert len(m2.__dict__["_mlp_pair"]) == 4
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
    m2.load_state_dict(m.state_dict())                      # the hook survived the round trip
    copy.deepcopy(m)


# --------------------------------------------------------------------------- GPU
def _module(d, params, dev):
    m = ga.MolecularGINConv(d).to(dev)
    m.load_state_dict({k: v.float() for k, v in params.items()})
    return m


def _check(dev, x, ei, ea, d, seed, frozen):
    """(1) the aggregation — the HIP part — forward and backward (d x, and the two embedding
    tables when they train) at the adapter's tolerance; (2) the whole layer's forward, and its
    d x on every node that no ReLU kink can reach: a hidden unit whose fp64 pre-activation is
    within 3e-5 of zero may switch sides in fp32 (the BLAS's, not this library's), which changes
    d agg of that node and hence d x of the node and its in-neighbours by O(1) — with ~10^6 hidden
    units per batch a handful of such units is expected, so those rows are left out."""
    params = oracle.init_params(d, seed=seed)
    gen = torch.Generator().manual_seed(seed + 1)
    g = torch.randn(x.shape, generator=gen)
    n = x.shape[0]
    ei_d, ea_d = ei.to(dev), ea.to(dev)

    # ---- (1) aggregation alone
    xo = x.double().requires_grad_(True)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    agg_o = oracle.gin_aggregate(xo, ei, ea, po)
    agg_o.backward(g.double())
    t = params["edge_embedding_type.weight"].float().to(dev).requires_grad_(not frozen)
    dr = params["edge_embedding_direction.weight"].float().to(dev).requires_grad_(not frozen)
    xd = x.to(dev).requires_grad_(True)
    agg = ga.gin.GinAggregateFunction.apply(xd, t, dr, ga.get_plan(ei_d, n, False),
                                            ga.gin.edge_code_counts(ei_d, ea_d, n))
    agg.backward(g.to(dev))
    # unnormalised sums over up to 80 neighbours with cancellation: absolute term relative to scale
    assert_close(agg.detach().cpu().double(), agg_o.detach(), "aggregate", 1e-5, grad_atol(agg_o.detach()))
    assert_close(xd.grad.cpu().double(), xo.grad, "d aggregate / d x", 1e-5, grad_atol(xo.grad))
    if not frozen:
        for got, k in ((t.grad, "edge_embedding_type.weight"), (dr.grad, "edge_embedding_direction.weight")):
            assert_close(got.cpu().double(), po[k].grad, f"grad {k}", 1e-5, grad_atol(po[k].grad))
    else:
        assert t.grad is None and dr.grad is None

    # ---- (2) the whole layer through the module
    want_out, want = oracle.gin_forward_backward(x.double(), ei, ea, params, g.double())
    m = _module(d, params, dev)
    if frozen:                                        # the adapter recipe: backbone frozen (gnn.py:230-241)
        for p in m.parameters():
            p.requires_grad_(False)
    xd = x.to(dev).requires_grad_(True)
    out = m(xd, ei_d, ea_d)
    out.backward(g.to(dev))
    # the MLP runs in the BLAS's fp32: dot products of 300-600 terms of O(1) -> relative to scale
    assert_close(out.detach().cpu().double(), want_out, "out", 1e-5, grad_atol(want_out, 2e-5))
    pre = torch.nn.functional.linear(agg_o.detach(), params["mlp.0.weight"], params["mlp.0.bias"])
    risky = (pre.abs() < 3e-5).any(1)
    reach = risky.clone()
    reach[ei[0][risky[ei[1]]]] = True                # d x[j] sees d agg[i] for every edge j -> i
    safe = ~reach
    assert safe.sum() >= 0.9 * n
    assert_close(xd.grad.cpu().double()[safe], want["x"][safe], "grad x", 1e-5, grad_atol(want["x"], 2e-5))
    if frozen:
        assert all(p.grad is None for p in m.parameters())
    else:
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("frozen", [True, False])
def test_gin_layer_esol_batch(dev, frozen):
    batch, cfg = ga.synth.make_workload("esol")
    _check(dev, batch.x, batch.edge_index, _bond_attr(batch.edge_index.shape[1], 5), cfg["d"], 0, frozen)


@pytest.mark.gpu
def test_gin_layer_hiv_shard(dev):
    batch, cfg = ga.synth.make_workload("hiv", graphs=128)
    _check(dev, batch.x, batch.edge_index, _bond_attr(batch.edge_index.shape[1], 6), cfg["d"], 1, True)


@pytest.mark.gpu
@pytest.mark.parametrize("frozen", [False, True])
@pytest.mark.parametrize("n,e,d", [(1, 0, 4), (9, 0, 8), (200, 1500, 64), (50, 4000, 128), (300, 900, 1024)])
def test_gin_layer_directed_graphs_with_duplicates_and_loops(dev, n, e, d, frozen):
    """frozen = True: the whole layer as ONE node (aggregation -> fp16 operand pair -> two GEMMs), at the
    widths where a row is narrower than one k-step of the GEMM (d = 4, 8) and at the widest supported."""
    g = torch.Generator().manual_seed(n + e)
    ei = torch.stack([torch.randint(n, (e,), generator=g), torch.randint(n, (e,), generator=g)])
    x = torch.randn(n, d, generator=g)
    _check(dev, x, ei, _bond_attr(e, 7), d, 2, frozen=frozen)


def _mlp_fn(scheme):
    return {"f16x2": ga.gin.GinPairMlpFunction, "bf16x3": ga.gin.GinMlpFunction}[scheme]


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("n,d", [(1, 8), (130, 64), (1, 300), (17, 300), (777, 300), (4099, 300), (13104, 300),
                                 (2000, 1024)])
def test_gin_mlp_is_fp32_accurate(dev, n, d, scheme):
    """The frozen layer's MLP on the MFMA kernels — round 4: csrc/mlp_f16.hip (every fp32 operand as two fp16
    values of a power-of-two scaled copy, three products per fp32 product), round 3: csrc/mlp.hip (three bf16
    values, six products), fp32 accumulation — against the fp64 MLP: forward and input gradient within
    rtol = atol = 1e-5 (atol relative to the tensor's scale, as for every sum of hundreds of O(1) terms
    here), and no worse than twice the error of torch's fp32 GEMMs on the same data — fp32-accurate, which
    is the gate for using it (GCONV_TORCH_MLP=1 is the A/B knob)."""
    gen = torch.Generator().manual_seed(n + d)
    layer = ga.MolecularGINConv(d).to(dev)
    for p in layer.parameters():
        p.requires_grad_(False)
    a = torch.randn(n, d, generator=gen)
    g = torch.randn(n, d, generator=gen)
    _mlp_case(dev, layer, a, g, scheme)


_TILED_GEMMS = r"""
import sys, torch
import gconv_adapter_amd as ga
from gconv_adapter_amd import gin
dev = torch.device("cuda:0")
n, d, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
gen = torch.Generator().manual_seed(n)
w1, b1 = torch.randn(2 * d, d, generator=gen).to(dev) * 0.1, torch.randn(2 * d, generator=gen).to(dev)
w2, b2 = torch.randn(d, 2 * d, generator=gen).to(dev) * 0.1, torch.randn(d, generator=gen).to(dev)
x, g = torch.randn(n, d, generator=gen).to(dev), torch.randn(n, d, generator=gen).to(dev)
ws = [gin._PairWeights() for _ in range(4)]
a2, g2 = gin._rows_pair(x), gin._rows_pair(g)
h2 = gin._pair_gemm(a2, n, d, ws[0].get(w1, b1, False), b1, None, 2 * d, 1, True)        # K = d, ReLU, pair out
y = gin._pair_gemm(h2, n, 2 * d, ws[1].get(w2, b2, False), b2, None, d, 0, False)        # K = 2d, fp32 out
gh2 = gin._pair_gemm(g2, n, d, ws[2].get(w2, None, True), None, h2, 2 * d, 2, True)      # K = d, masked, pair out
gx = gin._pair_gemm(gh2, n, 2 * d, ws[3].get(w1, None, True), None, None, d, 0, False)   # K = 2d, fp32 out
lib = ga._lib.load()
def defined(pair, f):   # the images [2][n][pad(f)] and the n row exponents (the rest of the buffer is padding)
    kp = (f + 31) // 32 * 32
    off = (2 * n * kp * 2 + 15) // 16 * 16
    return torch.cat([pair[:2 * n * kp * 2], pair[off:off + 4 * n]]).cpu()
torch.save({"h2": defined(h2, 2 * d), "y": y.cpu(), "gh2": defined(gh2, 2 * d), "gx": gx.cpu()}, out)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 530, 13104])
def test_gin_mlp_w_stationary_kernel_equals_the_tiled_kernel_bit_for_bit(dev, n, tmp_path):
    """csrc/mlp_f16.hip has two kernels for the same product: 128 x 80 tiles with k-steps through LDS, and (d = 300:
    K = 300 / 600) the weights stationary in registers with 16-row blocks of A streaming past them. Same
    products, same order per output element: the four GEMMs of a layer (forward and input gradient; fp32 and
    pair outputs, exponents included) agree bit for bit. Each variant runs in its own process (GCONV_MLP_WS is
    read once)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for ws in ("1", "0"):
        path = str(tmp_path / f"ws{ws}.pt")
        env = dict(os.environ, GCONV_MLP_WS=ws, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        subprocess.run([sys.executable, "-c", _TILED_GEMMS, str(n), "300", path], check=True, env=env, cwd=root,
                       timeout=300)
        outs[ws] = torch.load(path)
    for key, a in outs["1"].items():
        assert torch.equal(a, outs["0"][key]), key


def _mlp_case(dev, layer, a, g, scheme, row_scaled=False):
    fn = _mlp_fn(scheme)
    w1, b1, w2, b2 = (layer.mlp[0].weight, layer.mlp[0].bias, layer.mlp[2].weight, layer.mlp[2].bias)
    a64 = a.double().requires_grad_(True)
    h64 = torch.relu(torch.nn.functional.linear(a64, w1.cpu().double(), b1.cpu().double()))
    y64 = torch.nn.functional.linear(h64, w2.cpu().double(), b2.cpu().double())
    y64.backward(g.double())
    ad = a.to(dev).requires_grad_(True)
    y = fn.apply(ad, layer)
    y.backward(g.to(dev))
    at = a.to(dev).requires_grad_(True)
    yt = layer.mlp(at)                                   # torch's fp32 GEMMs on the same data
    yt.backward(g.to(dev))
    # a hidden unit within 3e-5 (of its row's scale) of the ReLU kink may switch sides in any fp32 evaluation:
    # leave the input-gradient rows of such nodes out (see _check)
    pre = torch.nn.functional.linear(a.double(), w1.cpu().double(), b1.cpu().double())
    safe = ~(pre.abs() < 3e-5 * pre.abs().amax(1, keepdim=True).clamp(min=1e-30 if row_scaled else 1.0)).any(1)
    for got, ref, want, what in ((y, yt, y64.detach(), "out"), (ad.grad[safe.to(dev)], at.grad[safe.to(dev)],
                                                                 a64.grad[safe], "d agg")):
        got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
        if got.numel() == 0:   # (n = 1 and its one node sits on a kink)
            continue
        if row_scaled:   # rows of very different magnitude: every row against its own scale
            scale = want.abs().amax(1, keepdim=True).clamp(min=1e-30)
            got, ref, want = got / scale, ref / scale, want / scale
        assert torch.isfinite(got).all(), what
        assert_close(got, want, what, 1e-5, grad_atol(want))
        if got.numel() >= 10000:   # (a maximum over a handful of rows says little: torch runs n = 1 as a GEMV)
            assert (got - want).abs().max() <= 2.0 * (ref - want).abs().max() + 1e-7, what
    assert torch.equal(fn.apply(ad, layer), y)     # bitwise repeatable


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tiny_rows", "huge_rows", "mixed_rows", "zero_rows", "big_bias", "big_weights",
                                  "one_hot_rows"])
def test_gin_mlp_f16x2_scaling_holds_over_the_fp32_range(dev, case):
    """The fp16 pair lives on its power-of-two scaling (one exponent per activation row, one per weight
    matrix, the hidden activation's from a bound): rows of magnitude 1e-12 .. 1e12 (gradients are tiny, some
    backbones' activations are not), rows that are exactly zero, a bias or weights far larger than the
    inputs — every row accurate relative to ITS OWN scale, nothing overflows fp16's 65504."""
    n, d = 640, 300
    gen = torch.Generator().manual_seed(len(case))
    layer = ga.MolecularGINConv(d).to(dev)
    for p in layer.parameters():
        p.requires_grad_(False)
    a = torch.randn(n, d, generator=gen)
    g = torch.randn(n, d, generator=gen)
    mag = torch.ones(n, 1)
    if case == "tiny_rows":
        mag = torch.full((n, 1), 1e-9)
    elif case == "huge_rows":
        mag = torch.full((n, 1), 1e9)
    elif case == "mixed_rows":
        mag = 10.0 ** torch.randint(-12, 13, (n, 1), generator=gen).float()
    elif case == "zero_rows":
        mag[::3] = 0.0
    elif case == "one_hot_rows":
        a = torch.zeros(n, d)
        a[torch.arange(n), torch.randint(0, d, (n,), generator=gen)] = 3.0
    with torch.no_grad():
        if case == "big_bias":
            layer.mlp[0].bias.mul_(1e4)
            layer.mlp[2].bias.mul_(1e4)
        if case == "big_weights":
            layer.mlp[0].weight.mul_(300.0)
            layer.mlp[2].weight.mul_(1e-3)
    layer.invalidate_mlp_split()
    # the forward's rows scale with `mag`; the gradient's rows get an independent spread
    gmag = 10.0 ** torch.randint(-10, 3, (n, 1), generator=gen).float() if case == "mixed_rows" else torch.ones(n, 1) * 1e-6
    _mlp_case(dev, layer, a * mag, g * gmag, "f16x2", row_scaled=True)


@pytest.mark.gpu
def test_frozen_layer_cpp_node_equals_the_python_node(dev):
    """The frozen layer's autograd node exists twice — csrc/torch_binding.cpp GinFrozenLayerFn (what the module
    uses when the binding is built) and gin.GinFrozenLayerFunction (ctypes) — around the same C ABI calls in
    the same order: output and input gradient agree bit for bit, also when the Python plan object is gone
    before the backward pass (the C++ node keeps its own copy of the CSR struct and the arena)."""
    if ga._torch_ext.get() is None:
        pytest.skip("the C++ binding is not built (or GCONV_PY_BINDING is set)")
    batch, cfg = ga.synth.make_workload("hiv", graphs=64)
    d, n = cfg["d"], batch.num_nodes
    gen = torch.Generator().manual_seed(3)
    e = batch.edge_index.shape[1]
    ea = torch.stack([torch.randint(0, 4, (e,), generator=gen), torch.randint(0, 3, (e,), generator=gen)], 1).to(dev)
    ei = batch.edge_index.to(dev)
    layer = ga.MolecularGINConv(d).to(dev)
    for p in layer.parameters():
        p.requires_grad_(False)
    go = torch.randn(n, d, generator=gen).to(dev)
    x1 = batch.x.to(dev).requires_grad_(True)
    y1 = layer(x1, ei, ea)                                    # the C++ node
    ga.PLAN_CACHE.clear()
    y1.backward(go)
    x2 = batch.x.to(dev).requires_grad_(True)
    plan = ga.get_plan(ei, n, False)
    counts = ga.gin.edge_code_counts(ei, ea, n, True)
    y2 = ga.gin.GinFrozenLayerFunction.apply(x2, plan, counts, layer)
    y2.backward(go)
    assert torch.equal(y1, y2) and torch.equal(x1.grad, x2.grad)
    with torch.no_grad():                                     # no graph: same numbers
        assert torch.equal(layer(x1.detach(), ei, ea), y1)


@pytest.mark.gpu
def test_gin_aggregate_alone_is_tight_and_deterministic(dev):
    """Without the MLP the HIP kernel is compared at the adapter's tolerance (1e-5) and bit-for-bit
    against itself; the code counts are exact integers."""
    batch, cfg = ga.synth.make_workload("hiv", graphs=64)
    d, n = cfg["d"], batch.num_nodes
    ea = _bond_attr(batch.edge_index.shape[1], 8)
    params = oracle.init_params(d, seed=3)
    want = oracle.gin_aggregate(batch.x.double(), batch.edge_index, ea, params)
    ei_d, ea_d = batch.edge_index.to(dev), ea.to(dev)
    counts = ga.gin.edge_code_counts(ei_d, ea_d, n)
    hist = np.zeros((n, 18), dtype=np.int64)
    np.add.at(hist, (batch.edge_index[1].numpy(), (ea[:, 0] * 3 + ea[:, 1]).numpy()), 1)
    np.testing.assert_array_equal(counts[:-1].view(n, 18).cpu().numpy(), hist)
    assert int(counts[-1]) == 0
    assert ga.gin.edge_code_counts(ei_d, ea_d, n) is counts          # one histogram per batch
    plan = ga.get_plan(ei_d, n, False)
    t = params["edge_embedding_type.weight"].float().to(dev)
    dr = params["edge_embedding_direction.weight"].float().to(dev)
    x = batch.x.to(dev)
    a = ga.gin.GinAggregateFunction.apply(x, t, dr, plan, counts)
    b = ga.gin.GinAggregateFunction.apply(x, t, dr, plan, counts)
    assert torch.equal(a, b)
    assert_close(a.cpu().double(), want, "aggregate")


@pytest.mark.gpu
def test_gin_shares_the_batch_plan_with_the_adapters_and_validates_attributes(dev):
    batch, cfg = ga.synth.make_workload("esol")
    d = cfg["d"]
    ei = batch.edge_index.to(dev)
    ea = _bond_attr(ei.shape[1], 9).to(dev)
    x = batch.x.to(dev)
    ga.PLAN_CACHE.clear()
    misses, hits = ga.PLAN_CACHE.misses, ga.PLAN_CACHE.hits
    adapter = ga.GConvAdapter(d, 4, normalize=False).to(dev)       # the reference's YAML default
    layer = ga.MolecularGINConv(d).to(dev)
    h = layer(adapter(x, ei, ea), ei, ea)
    assert h.shape == x.shape and ga.PLAN_CACHE.misses == misses + 1 and ga.PLAN_CACHE.hits == hits + 1
    bad = ea.clone()
    bad[3, 0] = 6                                                   # nn.Embedding would raise here
    out = layer(x, ei, bad)                                         # default: skipped, never a device fault
    assert torch.isfinite(out).all()
    layer.validate_edge_attr = True
    bad2 = bad.clone()
    with pytest.raises(RuntimeError, match="bond"):
        layer(x, ei, bad2)
    with pytest.raises(RuntimeError, match="edge_attr must be int64"):
        layer(x, ei, ea.float())


# ---------------------------------------------------------------------------
# NodeFormer's relational bias on the adapter's SpMM
# ---------------------------------------------------------------------------
def test_relational_bias_restatement_matches_dense_form():
    """CPU: oracle/nodeformer_rb_restated.py (index_add form of nodeformer.py:202-224) against the
    dense matrix D_in^-1/2 A_in D_out^-1/2 with duplicates, a loop and a directed edge."""
    from oracle import nodeformer_rb_restated as rb
    n = 7
    ei = torch.tensor([[0, 1, 1, 2, 3, 3, 5, 6, 0], [1, 0, 2, 1, 3, 4, 5, 6, 1]])
    x = torch.randn(1, n, 2, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    b = torch.tensor([0.3, -0.2], dtype=torch.float64)
    out = rb.add_conv_relational_bias(x, ei, b)
    row, col = ei
    din = torch.bincount(col, minlength=n).double()
    dout = torch.bincount(row, minlength=n).double()
    a = torch.zeros(n, n, dtype=torch.float64)
    for r, c in zip(row.tolist(), col.tolist()):
        a[c, r] += (1 / din[c]).sqrt() * (1 / dout[r]).sqrt()
    for i in range(2):
        assert torch.allclose(out[0, :, i], torch.sigmoid(b[i]) * a @ x[0, :, i], atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("heads,d,trans,kind", [(1, 32, "sigmoid", "adjs0"), (2, 32, "sigmoid", "adjs0"),
                                                (4, 32, "identity", "adjs1"), (1, 16, "sigmoid", "directed")])
def test_relational_bias_matches_restatement(dev, heads, d, trans, kind):
    """ga.add_conv_relational_bias (row scaling, ONE unnormalised CSR product, row scaling) against the
    restated reference function: output, d x and d b — on the adjacencies the transductive script
    builds (`finetune_transductive_learning.py:110-119`: loops re-added, then its square's pattern)
    and on a directed list with duplicates (separate in / out degrees)."""
    import gconv_adapter_amd as ga
    from oracle import nodeformer_rb_restated as rb
    g = torch.Generator().manual_seed(3)
    n = 600
    if kind == "directed":
        ei = torch.stack([torch.randint(n, (4000,), generator=g), torch.randint(n, (4000,), generator=g)])
    else:
        u = torch.randint(n, (1500,), generator=g); v = torch.randint(n, (1500,), generator=g)
        keep = u != v
        und = torch.unique(torch.cat([torch.stack([u[keep], v[keep]]), torch.stack([v[keep], u[keep]])], 1), dim=1)
        loops = torch.arange(n).repeat(2, 1)
        ei = torch.cat([und, loops], 1)
        if kind == "adjs1":   # pattern of the square (adj_mul, data_utils.py:208-213)
            a = torch.zeros(n, n); a[ei[0], ei[1]] = 1
            ei = (a @ a).nonzero().t().contiguous()
    x = torch.randn(1, n, heads, d, generator=g)
    b = torch.randn(heads, generator=g) * 0.5
    gout = torch.randn(1, n, heads, d, generator=g)
    x64 = x.double().requires_grad_(True); b64 = b.double().requires_grad_(True)
    want = rb.add_conv_relational_bias(x64, ei, b64, trans)
    want.backward(gout.double())
    xd = x.to(dev).requires_grad_(True); bd = b.to(dev).requires_grad_(True)
    got = ga.add_conv_relational_bias(xd, ei.to(dev), bd, trans)
    got.backward(gout.to(dev))
    def close(a, w, what, atol=1e-5):
        err = (a.detach().cpu().double() - w).abs()
        assert bool((err <= atol + 1e-5 * w.abs()).all()), f"{what}: max err {err.max().item():.3e}"
    close(got, want.detach(), "out")
    close(xd.grad, x64.grad, "dx")
    close(bd.grad, b64.grad, "db", atol=1e-5 * max(1.0, float(b64.grad.abs().max())))


@pytest.mark.gpu
def test_frozen_layer_built_directly_on_the_device(dev):
    """A layer constructed under `torch.device("cuda")` and never moved (no `_apply`, no `load_state_dict`)
    runs its frozen forward and backward (round-4 advisor finding: AttributeError on `_mlp_pair`)."""
    batch, cfg = ga.synth.make_workload("esol")
    d = cfg["d"]
    ei = batch.edge_index.to(dev)
    ea = _bond_attr(ei.shape[1], 5).to(dev)
    torch.manual_seed(0)
    with torch.device(dev):
        layer = ga.MolecularGINConv(d)
    moved = ga.MolecularGINConv(d)
    moved.load_state_dict({k: v.cpu() for k, v in layer.state_dict().items()})
    moved = moved.to(dev)
    for m in (layer, moved):
        for p in m.parameters():
            p.requires_grad_(False)
    x = batch.x.to(dev).requires_grad_(True)
    y = layer(x, ei, ea)
    y.square().sum().backward()
    x2 = batch.x.to(dev).requires_grad_(True)
    y2 = moved(x2, ei, ea)
    y2.square().sum().backward()
    assert torch.equal(y, y2) and torch.equal(x.grad, x2.grad)


@pytest.mark.gpu
def test_frozen_weight_images_follow_the_weights(dev):
    """The frozen MLP multiplies cached fp16 / bf16 images of its weights. `load_state_dict` and `.to()`
    rebuild them (hooks); a write through `.data`, which no version counter sees, needs
    `invalidate_mlp_split()` (round-3 advisor finding: it silently kept the stale images)."""
    batch, cfg = ga.synth.make_workload("esol")
    d = cfg["d"]
    ei = batch.edge_index.to(dev)
    ea = _bond_attr(ei.shape[1], 11).to(dev)
    layer = ga.MolecularGINConv(d).to(dev)
    for p in layer.parameters():
        p.requires_grad_(False)
    x = batch.x.to(dev)

    def want():
        agg = ga.gin.GinAggregateFunction.apply(x, layer.edge_embedding_type.weight, layer.edge_embedding_direction.weight,
                                                ga.get_plan(ei, x.shape[0], False), ga.gin.edge_code_counts(ei, ea, x.shape[0]))
        return layer.mlp(agg)

    def close(a, b):
        return torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
    y1 = layer(x, ei, ea)
    assert close(y1, want())
    other = ga.MolecularGINConv(d).state_dict()
    layer.load_state_dict(other)                      # hook: images rebuilt
    y2 = layer(x, ei, ea)
    assert close(y2, want()) and not close(y2, y1)
    layer.mlp[0].weight.data.mul_(2.0)                # invisible to the cache key ...
    assert torch.equal(layer(x, ei, ea), y2)          # ... so the stale images are still in use
    layer.invalidate_mlp_split()
    y3 = layer(x, ei, ea)
    assert close(y3, want()) and not close(y3, y2)
