"""torch.ops.gconv.* (TORCH_LIBRARY(gconv, ...) in csrc/torch_binding.cpp): the custom ops
compute what the nn.Module computes — same C ABI entry points — and `graph_prep` returns PyG's
gcn_norm structure as a CSR."""
import numpy as np
import pytest
import torch

import gconv_adapter_amd as ga
from oracle import pyg_restated as oracle
from tests.helpers import assert_close, grad_atol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    if ga._torch_ext.get() is None:
        pytest.skip("the torch custom ops live in the C++ binding (not built, or GCONV_PY_BINDING is set)")
    return ga.register_ops()


def _graph(n, e, seed, symmetric=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.stack([torch.randint(n, (e,), generator=g), torch.randint(n, (e,), generator=g)])
    return torch.cat([ei, ei.flip(0)], dim=1) if symmetric else ei


def test_ops_are_registered_with_the_surveyed_schemas(ops):
    names = {"graph_plan", "graph_prep", "adapter_fwd", "adapter_bwd", "adapter", "spmm"}
    for name in names:
        assert hasattr(ops, name), name
    assert str(ops.graph_prep.default._schema).endswith(
        "-> (Tensor rowptr, Tensor col, Tensor dis, bool symmetric, Tensor rowptr_t, Tensor col_t)")


@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("symmetric", [True, False])
def test_graph_prep_op_matches_numpy(dev, ops, normalize, symmetric):
    n = 700
    ei = _graph(n, 1500, 3, symmetric)
    rowptr, col, dis, sym, rowptr_t, col_t = ops.graph_prep(ei.to(dev), n, normalize)
    src, dst = ei.numpy()
    if normalize:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    order = np.lexsort((src, dst))
    want_rp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(want_rp, dst + 1, 1)
    want_rp = np.cumsum(want_rp)
    np.testing.assert_array_equal(rowptr.cpu().numpy(), want_rp)
    np.testing.assert_array_equal(col.cpu().numpy(), src[order])
    assert sym == symmetric
    if normalize:
        deg = np.diff(want_rp) + 1
        np.testing.assert_allclose(dis.cpu().numpy(), 1.0 / np.sqrt(deg), rtol=1e-6)
    else:
        assert dis.numel() == 0
    if symmetric:
        assert rowptr_t.numel() == 0 and col_t.numel() == 0
    else:
        order_t = np.lexsort((dst, src))
        np.testing.assert_array_equal(col_t.cpu().numpy(), dst[order_t])


@pytest.mark.parametrize("act", ["relu", "silu", "none"])
def test_adapter_ops_match_the_oracle(dev, ops, act):
    n, d, r = 900, 64, 8
    flags = oracle.AdapterFlags(non_linearity=act, normalization="none", learnable_scalar=True, skip_connection=True,
                                normalize=True)
    ei = _graph(n, 2000, 5)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(1))
    g = torch.randn(n, d, generator=torch.Generator().manual_seed(2))
    params = {k: v.float().double() for k, v in oracle.init_params(d, r, flags, seed=4).items()}
    want_out, want = oracle.adapter_forward_backward(x.double(), ei, params, flags, g.double(), need_dx=True)
    p = {k: v.float().to(dev) for k, v in params.items()}
    arena, plan = ops.graph_plan(ei.to(dev), n, True)
    args = (p["conv_down.lin.weight"], p["conv_down.bias"], p["conv_up.lin.weight"], p["conv_up.bias"], p["scalar"])
    out, saved = ops.adapter_fwd(x.to(dev), *args, arena, plan, act, True)
    assert_close(out.cpu().double(), want_out, "out")
    dx, dwd, dbd, dwu, dbu, ds = ops.adapter_bwd(g.to(dev), x.to(dev), saved, args[0], args[2], args[3], args[4], arena,
                                                 plan, act, True, True)
    got = {"x": dx, "conv_down.lin.weight": dwd, "conv_down.bias": dbd, "conv_up.lin.weight": dwu,
           "conv_up.bias": dbu, "scalar": ds}
    for k, w in want.items():
        assert_close(got[k].cpu().double(), w, f"grad {k}", atol=1e-5 if k == "x" else grad_atol(w))
    # the differentiable op: same numbers through autograd
    xs = x.to(dev).requires_grad_(True)
    leaves = [a.clone().requires_grad_(True) for a in args]
    out2 = ops.adapter(xs, *leaves, arena, plan, act, True)
    assert torch.equal(out2, out)
    out2.backward(g.to(dev))
    assert torch.equal(xs.grad, dx)
    for leaf, ref in zip(leaves, (dwd, dbd, dwu, dbu, ds)):
        assert torch.equal(leaf.grad, ref)


def test_spmm_op(dev, ops):
    n = 500
    ei = _graph(n, 1200, 9, symmetric=False)
    h = torch.randn(n, 16, generator=torch.Generator().manual_seed(3))
    arena, plan = ops.graph_plan(ei.to(dev), n, False)
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((ei[1], ei[0]), torch.ones(ei.shape[1], dtype=torch.float64), accumulate=True)
    assert_close(ops.spmm(arena, plan, h.to(dev), False).cpu().double(), a @ h.double(), "S h")
    assert_close(ops.spmm(arena, plan, h.to(dev), True).cpu().double(), a.t() @ h.double(), "S^T h")


def test_ops_raise_runtime_errors(dev, ops):
    ei = _graph(50, 100, 1).to(dev)
    arena, plan = ops.graph_plan(ei, 50, True)
    with pytest.raises(RuntimeError, match="graph plan was built for"):
        ops.adapter_fwd(torch.randn(40, 32, device=dev), torch.randn(4, 32, device=dev), None,
                        torch.randn(32, 4, device=dev), None, None, arena, plan, "relu", True)
    with pytest.raises(RuntimeError, match="non_linearity"):
        ops.adapter_fwd(torch.randn(50, 32, device=dev), torch.randn(4, 32, device=dev), None,
                        torch.randn(32, 4, device=dev), None, None, arena, plan, "gelu", True)
    bad = ei.clone(); bad[0, 3] = 77
    with pytest.raises(RuntimeError, match="outside"):
        ops.graph_plan(bad, 50, True)
    # parameter tensors are read through raw pointers: a CPU weight, a short bias, a wrong dtype must be an
    # error, never a device fault or an out-of-bounds read (round-3 advisor finding)
    x, wd, wu = torch.randn(50, 32, device=dev), torch.randn(4, 32, device=dev), torch.randn(32, 4, device=dev)
    with pytest.raises(RuntimeError, match="conv_down.lin.weight"):
        ops.adapter_fwd(x, wd.cpu(), None, wu, None, None, arena, plan, "relu", True)
    with pytest.raises(RuntimeError, match="conv_up.bias"):
        ops.adapter_fwd(x, wd, None, wu, torch.zeros(31, device=dev), None, arena, plan, "relu", True)
    with pytest.raises(RuntimeError, match="conv_down.bias"):
        ops.adapter_fwd(x, wd, torch.zeros(4, device=dev, dtype=torch.float64), wu, None, None, arena, plan, "relu", True)
    with pytest.raises(RuntimeError, match="scalar"):
        ops.adapter(x, wd, None, wu, None, torch.ones(1), arena, plan, "relu", True)
    out, saved = ops.adapter_fwd(x, wd, None, wu, None, None, arena, plan, "relu", True)
    with pytest.raises(RuntimeError, match="conv_up.bias"):
        ops.adapter_bwd(torch.randn_like(x), x, saved, wd, wu, torch.zeros(32), None, arena, plan, "relu", True, True)
