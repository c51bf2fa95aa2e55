"""CPU tests of the oracle itself (`-m "not gpu"`): the two independent derivations
agree, autograd agrees with the hand-written backward, the restated PyG semantics
hold, and the committed golden fixtures are what the
oracle produces."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import dense_ref
from oracle import pyg_restated as oracle

from helpers import golden_case, load_golden


def _rand_graph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randint(n, (e,), generator=g), torch.randint(n, (e,), generator=g)])


def _np_params(p):
    return {k: v.numpy() for k, v in p.items()}


@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("act", ["relu", "silu", "none"])
@pytest.mark.parametrize("skip,scalar", [(True, True), (False, False), (True, False)])
def test_edge_list_form_matches_dense_form(normalize, act, skip, scalar):
    n, d, r = 23, 12, 5
    ei = _rand_graph(n, 90, seed=3)
    flags = oracle.AdapterFlags(non_linearity=act, learnable_scalar=scalar, skip_connection=skip,
                                normalize=normalize)
    params = oracle.init_params(d, r, flags, seed=4)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, d, generator=g, dtype=torch.float64)
    gout = torch.randn(n, d, generator=g, dtype=torch.float64)
    out, grads = oracle.adapter_forward_backward(x, ei, params, flags, gout)
    out2, grads2 = dense_ref.forward_backward(x.numpy(), ei.numpy(), _np_params(params), gout.numpy(),
                                              normalize=normalize, non_linearity=act,
                                              skip_connection=skip, learnable_scalar=scalar)
    np.testing.assert_allclose(out.numpy(), out2, rtol=1e-11, atol=1e-11)
    for k, v in grads2.items():
        np.testing.assert_allclose(grads[k].numpy(), v, rtol=1e-10, atol=1e-10, err_msg=k)


def test_reassociated_form_equals_pyg_order():
    n, d, r = 40, 16, 4
    ei = _rand_graph(n, 150, seed=7)
    for normalize in (True, False):
        flags = oracle.AdapterFlags(learnable_scalar=True, normalize=normalize)
        params = oracle.init_params(d, r, flags, seed=8)
        x = torch.randn(n, d, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
        a = oracle.adapter_forward(x, ei, params, flags)
        b = oracle.adapter_forward_reassociated(x, ei, params, flags)
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)


# ---- restated PyG semantics, one named test each ----------------------
def test_sem1_direction_source_to_target():
    h = torch.tensor([[1.0], [10.0], [100.0]], dtype=torch.float64)
    ei = torch.tensor([[0], [2]])                     # 0 -> 2
    y = oracle.propagate(h, ei, None, 3)
    assert y.flatten().tolist() == [0.0, 0.0, 1.0]


def test_sem2_duplicate_edges_count_twice():
    h = torch.tensor([[1.0], [10.0]], dtype=torch.float64)
    ei = torch.tensor([[0, 0], [1, 1]])
    assert oracle.propagate(h, ei, None, 2)[1].item() == 2.0


def test_sem3_unnormalized_has_no_self_loops_and_isolated_row_is_bias():
    x = torch.eye(3, dtype=torch.float64)
    w = torch.eye(3, dtype=torch.float64)
    b = torch.tensor([0.5, 0.5, 0.5], dtype=torch.float64)
    y = oracle.gcn_conv(x, torch.tensor([[0], [1]]), w, b, normalize=False)
    assert torch.equal(y[2], b) and torch.equal(y[0], b)          # no loop, no in-edge
    assert torch.equal(y[1], b + x[0])


def test_sem4_normalize_drops_existing_loops_and_adds_exactly_one():
    ei = torch.tensor([[0, 0, 1, 1], [0, 0, 1, 0]])                # two loops on 0, one on 1, 1->0
    e2, w = oracle.gcn_norm(ei, 2, torch.float64)
    pairs = sorted(zip(e2[0].tolist(), e2[1].tolist()))
    assert pairs == [(0, 0), (1, 0), (1, 1)]
    # in-degree on target of the looped graph: deg(0) = 2, deg(1) = 1
    S = dense_ref.propagation_matrix(ei.numpy(), 2, True)
    np.testing.assert_allclose(S, [[0.5, 2 ** -0.5], [0.0, 1.0]])


def test_sem5_linear_before_propagate_bias_after():
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0]], dtype=torch.float64)
    w = torch.tensor([[1.0, 1.0]], dtype=torch.float64)
    b = torch.tensor([7.0], dtype=torch.float64)
    y = oracle.gcn_conv(x, torch.tensor([[0, 1], [1, 1]]), w, b, normalize=False)
    assert y.flatten().tolist() == [7.0, 3.0 + 7.0 + 7.0]         # bias added once, after aggregation


def test_sem6_node_dim_is_minus_two():
    n, d, r = 9, 8, 3
    ei = _rand_graph(n, 30, 1)
    flags = oracle.AdapterFlags(learnable_scalar=True)
    params = oracle.init_params(d, r, flags)
    x = torch.randn(n, d, dtype=torch.float64)
    a = oracle.adapter_forward(x, ei, params, flags)
    b = oracle.adapter_forward(x.unsqueeze(0), ei, params, flags)
    assert b.shape == (1, n, d) and torch.allclose(a, b[0])


def test_sem7_batchnorm_skipped_when_leading_dim_is_one():
    n, d, r = 9, 8, 3
    ei = _rand_graph(n, 30, 1)
    fl_bn = oracle.AdapterFlags(normalization="batch_norm")
    fl_no = oracle.AdapterFlags(normalization="none")
    params = oracle.init_params(d, r, fl_bn)
    x = torch.randn(1, n, d, dtype=torch.float64)
    assert torch.equal(oracle.adapter_forward(x, ei, params, fl_bn), oracle.adapter_forward(x, ei, params, fl_no))


def test_sem8_residual_before_norm_before_scalar():
    n, d, r = 6, 4, 2
    ei = _rand_graph(n, 12, 2)
    flags = oracle.AdapterFlags(normalization="layer_norm", learnable_scalar=True)
    params = oracle.init_params(d, r, flags)
    x = torch.randn(n, d, dtype=torch.float64)
    base = oracle.AdapterFlags(normalization="none", learnable_scalar=False)
    pre = oracle.adapter_forward(x, ei, params, base)              # conv + x
    want = torch.nn.functional.layer_norm(pre, (d,), params["normalization.weight"],
                                          params["normalization.bias"]) * params["scalar"]
    assert torch.allclose(oracle.adapter_forward(x, ei, params, flags), want)


def test_sem9_batch_norm_tail_follows_the_module_training_flag():
    """`training` = the reference module's `.training`: its `nn.BatchNorm1d` (`gconv_adapter.py:54-55,98-100`)
    normalises with batch statistics and updates its running buffers under `model.train()`
    (`finetune_inductive_learning.py:41`), and with the running buffers under `model.eval()` (`:96`). The
    restatement is checked against that very class in fp64."""
    n, d, r = 40, 8, 3
    fl = oracle.AdapterFlags(normalization="batch_norm", learnable_scalar=True)
    base = oracle.AdapterFlags(normalization="none", learnable_scalar=False)
    params = oracle.init_params(d, r, fl)
    params["normalization.running_mean"] = torch.zeros(d, dtype=torch.float64)
    params["normalization.running_var"] = torch.ones(d, dtype=torch.float64)
    bn = torch.nn.BatchNorm1d(d).double()
    with torch.no_grad():
        bn.weight.copy_(params["normalization.weight"]); bn.bias.copy_(params["normalization.bias"])
    for step in range(3):                                   # three training steps: buffers move together
        ei = _rand_graph(n, 120, 10 + step)
        x = torch.randn(n, d, dtype=torch.float64, generator=torch.Generator().manual_seed(step)) + step
        pre = oracle.adapter_forward(x, ei, params, base)   # conv + x
        bn.train()
        want = bn(pre) * params["scalar"]
        got = oracle.adapter_forward(x, ei, params, fl, training=True)
        assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)
        assert torch.allclose(params["normalization.running_mean"], bn.running_mean, rtol=1e-12, atol=1e-12)
        assert torch.allclose(params["normalization.running_var"], bn.running_var, rtol=1e-12, atol=1e-12)
    ei = _rand_graph(n, 100, 99)
    x = torch.randn(n, d, dtype=torch.float64)
    bn.eval()
    frozen = {k: v.clone() for k, v in params.items()}
    want = bn(oracle.adapter_forward(x, ei, params, base)) * params["scalar"]
    got = oracle.adapter_forward(x, ei, params, fl, training=False)
    assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)
    scratch = {k: v.clone() for k, v in params.items()}     # (a training call updates its buffers in place)
    assert not torch.allclose(got, oracle.adapter_forward(x, ei, scratch, fl, training=True))   # (it matters)
    for k in ("normalization.running_mean", "normalization.running_var"):   # evaluation does not touch the buffers
        assert torch.equal(params[k], frozen[k])
    with pytest.raises(ValueError):                         # nothing to evaluate with
        oracle.adapter_forward(x, ei, {k: v for k, v in params.items() if "running" not in k}, fl, training=False)
    # without buffers training mode is what every fixture used so far
    bare = {k: v for k, v in params.items() if "running" not in k}
    assert torch.allclose(oracle.adapter_forward(x, ei, bare, fl), oracle.adapter_forward(x, ei, bare, fl, training=True))


def test_invalid_flags_raise_value_error():
    x = torch.randn(3, 4, dtype=torch.float64)
    ei = torch.tensor([[0], [1]])
    p = oracle.init_params(4, 2, oracle.AdapterFlags())
    with pytest.raises(ValueError):
        oracle.adapter_forward(x, ei, p, oracle.AdapterFlags(non_linearity="gelu"))
    with pytest.raises(ValueError):
        oracle.adapter_forward(x, ei, p, oracle.AdapterFlags(normalization="group_norm"))


# ---- golden fixtures are reproducible from the oracle ----------------------------------
@pytest.mark.parametrize("name", ["g1_path5_norm.npz", "g1_path5_raw.npz", "g2_molecules.npz",
                                  "g3_transductive.npz"])
def test_golden_fixture_matches_oracle(name):
    case = golden_case(load_golden(name))
    params = {k: v.double() for k, v in case["params"].items()}
    out, grads = oracle.adapter_forward_backward(case["x"].double(), case["edge_index"], params,
                                                 case["flags"], case["grad_out"].double())
    assert torch.allclose(out, case["expected"]["out"], rtol=1e-12, atol=1e-12)
    for k, v in case["expected"].items():
        if k != "out":
            assert torch.allclose(grads[k], v, rtol=1e-12, atol=1e-12), k


def test_golden_flag_grid_matches_oracle_and_dense_form():
    rec = load_golden("g4_flags.npz")
    assert rec["x"].shape[0] == 36
    for i in range(36):
        case = golden_case(rec, i)
        params = {k: v.double() for k, v in case["params"].items()}
        out, grads = oracle.adapter_forward_backward(case["x"].double(), case["edge_index"], params,
                                                     case["flags"], case["grad_out"].double())
        assert torch.allclose(out, case["expected"]["out"], rtol=1e-12, atol=1e-12)
        f = case["flags"]
        if f.normalization == "none":   # the dense form covers the norm-free composition
            out2, grads2 = dense_ref.forward_backward(
                case["x"].double().numpy(), case["edge_index"].numpy(), _np_params(params),
                case["grad_out"].double().numpy(), normalize=f.normalize,
                non_linearity=f.non_linearity, skip_connection=f.skip_connection,
                learnable_scalar=f.learnable_scalar)
            np.testing.assert_allclose(out2, case["expected"]["out"].numpy(), rtol=1e-10, atol=1e-10)
            for k, v in grads2.items():
                np.testing.assert_allclose(v, case["expected"][k].numpy(), rtol=1e-9, atol=1e-9)


def test_g1_fixture_has_the_documented_edge_cases():
    case = golden_case(load_golden("g1_path5_norm.npz"))
    ei = case["edge_index"]
    pairs = list(zip(ei[0].tolist(), ei[1].tolist()))
    assert (2, 2) in pairs                                   # pre-existing self-loop
    assert pairs.count((1, 2)) == 2                          # duplicate edge
    assert (3, 0) in pairs and (0, 3) not in pairs           # directed-only edge
    assert 4 not in ei.flatten().tolist()                    # isolated node


# ---- third derivation: plain C, hand-written backward in PyG's op order ------------------
def _c_oracle():
    from oracle import c_oracle
    c_oracle.load()
    return c_oracle


@pytest.mark.parametrize("name", ["g1_path5_norm.npz", "g1_path5_raw.npz", "g2_molecules.npz"])
def test_c_oracle_matches_golden_fixtures(name):
    """oracle/gconv_oracle.c (scalar loops over the edge list, conv_up propagated at width d as
    PyG does, backward written by hand) reproduces the committed golden vectors."""
    c = _c_oracle()
    case = golden_case(load_golden(name))
    params = {k: v.double() for k, v in case["params"].items()}
    out, grads = c.adapter_forward_backward(case["x"].double(), case["edge_index"], params, case["flags"],
                                            case["grad_out"].double())
    assert torch.allclose(out, case["expected"]["out"], rtol=1e-11, atol=1e-11)
    for k, v in case["expected"].items():
        if k != "out":
            assert torch.allclose(grads[k], v, rtol=1e-10, atol=1e-10), k


def test_c_oracle_matches_golden_flag_grid():
    c = _c_oracle()
    rec = load_golden("g4_flags.npz")
    seen = 0
    for i in range(36):
        case = golden_case(rec, i)
        if case["flags"].normalization != "none":
            continue
        seen += 1
        params = {k: v.double() for k, v in case["params"].items()}
        out, grads = c.adapter_forward_backward(case["x"].double(), case["edge_index"], params, case["flags"],
                                                case["grad_out"].double())
        assert torch.allclose(out, case["expected"]["out"], rtol=1e-11, atol=1e-11)
        for k, v in case["expected"].items():
            if k != "out":
                assert torch.allclose(grads[k], v, rtol=1e-10, atol=1e-10), (i, k)
    assert seen == 12


@pytest.mark.parametrize("seed", range(6))
def test_c_oracle_matches_torch_oracle_on_random_graphs(seed):
    """directed lists with duplicates and self-loops, isolated nodes, both normalisation modes"""
    c = _c_oracle()
    g = torch.Generator().manual_seed(100 + seed)
    n, d, r = 30 + 7 * seed, 8 + 4 * seed, 1 + seed
    ei = _rand_graph(n - 3, 4 * n, seed=200 + seed)          # the last 3 nodes stay isolated
    flags = oracle.AdapterFlags(non_linearity=["relu", "silu", "none"][seed % 3], learnable_scalar=seed % 2 == 0,
                                skip_connection=seed % 3 != 1, normalize=seed % 2 == 1)
    params = oracle.init_params(d, r, flags, seed=300 + seed)
    x = torch.randn(n, d, generator=g, dtype=torch.float64)
    gout = torch.randn(n, d, generator=g, dtype=torch.float64)
    out, grads = oracle.adapter_forward_backward(x, ei, params, flags, gout)
    out2, grads2 = c.adapter_forward_backward(x, ei, params, flags, gout)
    assert torch.allclose(out, out2, rtol=1e-11, atol=1e-11)
    for k, v in grads2.items():
        assert torch.allclose(grads[k], v, rtol=1e-10, atol=1e-10), k


def test_c_oracle_rejects_bad_edges():
    c = _c_oracle()
    flags = oracle.AdapterFlags()
    params = oracle.init_params(4, 2, flags)
    x = torch.zeros(3, 4, dtype=torch.float64)
    with pytest.raises(RuntimeError):
        c.adapter_forward_backward(x, torch.tensor([[0, 5], [1, 1]]), params, flags, x)
