"""Readout and loss at the end of the fine-tune step: csrc/readout.hip behind
``ga.global_mean_pool`` / ``ga.global_add_pool`` / ``ga.masked_loss`` against ``oracle/readout_restated.py``."""
from __future__ import annotations

import math

import pytest
import torch

import gconv_adapter_amd as ga
from oracle import readout_restated as oracle

from helpers import assert_close


# ---- the oracle itself (CPU) -------------------------------------------------------------------------------------
def test_oracle_pool_is_a_scatter_over_the_batch_vector():
    x = torch.arange(12, dtype=torch.float64).view(6, 2)
    batch = torch.tensor([0, 0, 2, 2, 2, 3])
    s = oracle.global_pool(x, batch, 5, mean=False)
    m = oracle.global_pool(x, batch, 5, mean=True)
    assert s.tolist() == [[2, 4], [0, 0], [18, 21], [10, 11], [0, 0]]          # graphs 1 and 4 have no nodes
    assert m.tolist() == [[1, 2], [0, 0], [6, 7], [10, 11], [0, 0]]


def test_oracle_masked_loss_follows_the_loop():
    out = torch.tensor([[0.5], [-2.0], [3.0], [0.0]])
    tgt = torch.tensor([[1.0], [float("nan")], [0.0], [1.0]], dtype=torch.float64)
    want = sum(math.log1p(math.exp(-x)) if t == 1 else x + math.log1p(math.exp(-x)) for x, t in ((0.5, 1), (3.0, 0), (0.0, 1))) / 3
    assert abs(float(oracle.masked_loss(out, tgt, "bce")) - want) < 1e-12
    assert abs(float(oracle.masked_loss(out, tgt, "mse")) - (0.25 + 9.0 + 1.0) / 3) < 1e-12
    assert float(oracle.masked_loss(out, torch.full((4, 1), float("nan"), dtype=torch.float64), "bce")) == 0.0


# ---- the kernels (GPU) -------------------------------------------------------------------------------------------
@pytest.fixture
def dev():
    return torch.device("cuda:0")


def _batch_vector(sizes):
    return torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))


@pytest.mark.gpu
@pytest.mark.parametrize("mean", [True, False])
@pytest.mark.parametrize("sizes,d", [([3, 1, 7, 2], 8), ([5], 300), ([0, 4, 0, 0, 9, 1, 0], 32),
                                     ([1] * 70 + [222] + [2] * 300, 300), ([40] * 9, 1024)])
def test_global_pool_matches_the_restatement(dev, sizes, d, mean):
    """Forward and input gradient against the scatter restatement in fp64 (rtol = atol = 1e-5: a graph's rows
    are added in node order in fp32), graphs without nodes included (zeros, no gradient), bitwise repeatable."""
    g = torch.Generator().manual_seed(len(sizes) + d)
    batch = _batch_vector(sizes)
    n = batch.shape[0]
    x = torch.randn(n, d, generator=g)
    go = torch.randn(len(sizes), d, generator=g)
    x64 = x.double().requires_grad_(True)
    want = oracle.global_pool(x64, batch, len(sizes), mean)
    want.backward(go.double())
    fn = ga.global_mean_pool if mean else ga.global_add_pool
    xd = x.to(dev).requires_grad_(True)
    got = fn(xd, batch.to(dev), len(sizes))
    got.backward(go.to(dev))
    assert_close(got.detach().cpu().double(), want.detach(), "pooled", 1e-5, 1e-5)
    assert_close(xd.grad.cpu().double(), x64.grad, "d x", 1e-5, 1e-5)
    assert torch.equal(fn(xd, batch.to(dev), len(sizes)), got)


@pytest.mark.gpu
def test_global_pool_on_a_molecule_batch_and_the_other_paths(dev):
    batch, cfg = ga.synth.make_workload("hiv")
    sizes = (batch.graph_ptr[1:] - batch.graph_ptr[:-1]).tolist()
    bv = _batch_vector(sizes).to(dev)
    x = batch.x.to(dev)
    want = oracle.global_pool(batch.x.double(), bv.cpu(), batch.num_graphs, True)
    assert_close(ga.global_mean_pool(x, bv, batch.num_graphs).cpu().double(), want, "hiv", 1e-5, 1e-5)
    assert_close(ga.global_mean_pool(x, bv).cpu().double(), want, "size=None", 1e-5, 1e-5)      # (reads batch.max())
    # one graph (PyG: batch=None), a width the kernel does not take (torch's scatter), an unsorted batch vector
    assert_close(ga.global_add_pool(x, None).cpu().double(), batch.x.double().sum(0, keepdim=True), "none", 1e-5, 1e-4)
    x6 = torch.randn(bv.shape[0], 6, device=dev)
    assert_close(ga.global_mean_pool(x6, bv, batch.num_graphs).cpu().double(),
                 oracle.global_pool(x6.cpu().double(), bv.cpu(), batch.num_graphs, True), "d=6", 1e-5, 1e-5)
    with pytest.raises(RuntimeError, match="sorted"):
        ga.global_mean_pool(x, bv.flip(0), batch.num_graphs, validate=True)
    # round-4 advisor finding: an unsorted batch vector (or ids outside [0, size)) used to give silently wrong sums.
    # The kernel now checks the order on its way: the affected graphs' rows are NaN (no host synchronisation needed
    # to notice: the loss is NaN), the sorted neighbours are untouched, and validate=True raises.
    swapped = bv.clone()
    lo, hi = int(batch.graph_ptr[7]), int(batch.graph_ptr[9])          # two nodes of graphs 7 and 8 trade places
    swapped[lo], swapped[hi - 1] = bv[hi - 1], bv[lo]
    got = ga.global_mean_pool(x, swapped, batch.num_graphs)
    bad = torch.isnan(got).any(1).cpu()
    assert (bad[7] or bad[8]) and not bad[:7].any() and not bad[9:].any()      # the disorder shows, and only there
    ok = torch.ones_like(bad); ok[7:9] = False
    assert_close(got[ok.to(dev)].cpu().double(), want[ok], "sorted graphs next to a disorder", 1e-5, 1e-5)
    with pytest.raises(RuntimeError, match="sorted"):
        ga.global_mean_pool(x, swapped, batch.num_graphs, validate=True)
    assert torch.isnan(ga.global_add_pool(x, bv, batch.num_graphs - 1)).any()     # ids >= size: flagged, not dropped
    with pytest.raises(RuntimeError, match="sorted"):
        ga.global_add_pool(x, bv, batch.num_graphs - 1, validate=True)
    assert not torch.isnan(ga.global_mean_pool(x, bv, batch.num_graphs, validate=True)).any()
    with pytest.raises(RuntimeError, match="int64"):
        ga.global_mean_pool(x, bv.int(), batch.num_graphs)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ga.global_mean_pool(batch.x, bv.cpu(), batch.num_graphs)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["bce", "mse"])
@pytest.mark.parametrize("shape,missing", [((512, 1), 0.1), ((32, 1), 0.0), ((4096, 12), 0.7), ((5, 3), 1.0), ((3000, 1), 0.5),
                                           ((4096, 617), 0.3), ((700, 24), 1.0)])   # (> 16384 logits: 256 workgroups, two launches)
def test_masked_loss_matches_the_loop(dev, kind, shape, missing):
    """Loss (fp64, rtol 1e-12: the sum runs in a different but fixed order) and d loss / d output (fp32 storage
    of an fp64 value: rtol 1e-6) against the restated loop, logits up to +-40 (no overflow in either tail)."""
    g = torch.Generator().manual_seed(shape[0] + shape[1])
    out = torch.randn(shape, generator=g) * 3
    out.view(-1)[:4] = torch.tensor([40.0, -40.0, 0.0, 17.5])[: min(4, out.numel())]
    tgt = torch.randint(0, 2, shape, generator=g).double() if kind == "bce" else torch.randn(shape, generator=g).double()
    tgt[torch.rand(shape, generator=g) < missing] = float("nan")
    o64 = out.clone().requires_grad_(True)
    want = oracle.masked_loss(o64, tgt, kind)
    want.backward()
    od = out.to(dev).requires_grad_(True)
    got = ga.masked_loss(od, tgt.to(dev), kind)
    assert got.dtype == torch.float64 and got.dim() == 0
    (got * 2.0).backward()                                   # (an upstream factor: the backward is a scaling)
    assert abs(float(got.detach()) - float(want.detach())) <= 1e-12 * max(1.0, abs(float(want.detach())))
    assert_close(od.grad.cpu().double(), 2.0 * o64.grad.double(), "d output", 1e-6, 1e-9)
    assert torch.equal(ga.masked_loss(od, tgt.to(dev), kind), got)


@pytest.mark.gpu
def test_masked_loss_arguments(dev):
    o = torch.randn(8, 1, device=dev)
    with pytest.raises(ValueError, match="kind"):
        ga.masked_loss(o, o.double(), "l1")
    with pytest.raises(RuntimeError, match="fp64"):
        ga.masked_loss(o, o, "bce")
    with pytest.raises(RuntimeError, match="no CPU path"):
        ga.masked_loss(o.cpu(), o.cpu().double(), "bce")
