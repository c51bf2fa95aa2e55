"""Round 5: the general SpMM's row -> XCD schedule for graphs with local neighbourhoods.

The radix builder counts the stored edges whose column lies in its row's eighth of the node range
(``gconv_csr_t::near_edges``); when they are at least half of nnz (and the graph has >= 32768 nodes) the row
kernel hands every XCD a contiguous eighth of the row blocks instead of every eighth block — with round-robin
blocks each of the eight private L2s fetched every H row on the banded control (``profiles/r05_spmm_banded_pmc.json``).
The schedule is a permutation of the row blocks: a row's sum does not depend on it. Counterpart of PyG's
``propagate`` at the reference's call site ``/root/reference/src/finetune/gconv_adapter.py:92``."""
from __future__ import annotations

import os
import subprocess
import sys

import pytest
import torch

import gconv_adapter_amd as ga
from oracle import pyg_restated as oracle

from helpers import assert_close, check_workload

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _banded(n, hidden, seed=0, band=512):
    b = ga.synth.banded_like(n, 8, band, hidden=hidden, seed=seed, with_x=True)
    return b


def test_locality_statistic_of_the_radix_builder(dev):
    n = 50_000
    b = _banded(n, 8)
    ei = b.edge_index
    plan = ga.GraphPlan(ei.to(dev), n, True)
    keep = ei[0] != ei[1]                       # (normalised: loops dropped)
    want = int(((ei[0][keep] * 8) // n == (ei[1][keep] * 8) // n).sum())
    assert plan.fwd.near_edges == want and 2 * want >= plan.fwd.nnz
    plan = ga.GraphPlan(ei.to(dev), n, False)   # unnormalised: every stored edge counts
    assert plan.fwd.near_edges == int(((ei[0] * 8) // n == (ei[1] * 8) // n).sum())
    r = ga.synth.arxiv_like(40_000, 150_000, hidden=8, seed=1)      # uniformly random columns: about an eighth
    rp = ga.GraphPlan(r.edge_index.to(dev), r.num_nodes, True)
    assert 0 < rp.fwd.near_edges < rp.fwd.nnz // 4
    g = torch.Generator().manual_seed(3)
    d = torch.stack([torch.randint(n, (90_000,), generator=g), torch.randint(n // 8, (90_000,), generator=g)])
    dp = ga.GraphPlan(d.to(dev), n, False)      # directed: both directions counted, each on its own CSR
    assert dp.fwd.near_edges == int(((d[0] * 8) // n == (d[1] * 8) // n).sum()) == dp.bwd.near_edges
    small = ga.synth.molecule_batch("esol", 32, hidden=8)           # the one-launch builders do not count
    assert ga.GraphPlan(small.edge_index.to(dev), small.num_nodes, True).fwd.near_edges == 0


@pytest.mark.parametrize("width,normalize", [(8, True), (32, True), (32, False), (64, True)])
def test_spmm_on_a_banded_graph_matches_the_oracle(dev, width, normalize):
    n = 50_011                                   # (row blocks not a multiple of 8: the tail keeps its place)
    b = _banded(n, 8, seed=width)
    plan = ga.GraphPlan(b.edge_index.to(dev), n, normalize)
    assert 2 * plan.fwd.near_edges >= plan.fwd.nnz
    h = torch.randn(n, width, generator=torch.Generator().manual_seed(2))
    y = ga.spmm(plan, h.to(dev))
    if normalize:
        e2, w = oracle.gcn_norm(b.edge_index, n, torch.float64)
        want = oracle.propagate(h.double(), e2, w, n)
    else:
        want = oracle.propagate(h.double(), b.edge_index, None, n)
    assert_close(y.cpu().double(), want, f"banded spmm F={width}")
    assert torch.equal(ga.spmm(plan, h.to(dev)), y)


def test_adapter_on_a_banded_graph_matches_the_oracle(dev):
    """Forward + backward through the four-launch general path (a band is not tileable): K2 / K2' / B2 / B3 under the
    XCD-contiguous schedule, incl. the bias-gradient partials whose slots are the (permuted) row blocks."""
    n, d, r = 40_000, 64, 16
    b = _banded(n, d, seed=5, band=300)
    plan = ga.get_plan(b.edge_index.to(dev), n, True)
    assert plan.n_tiles == 0 and 2 * plan.fwd.near_edges >= plan.fwd.nnz
    flags = oracle.AdapterFlags(non_linearity="silu", learnable_scalar=True)
    check_workload(ga, dev, b, d, r, flags, seed=3)


_AB_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
n = 50_011
b = ga.synth.banded_like(n, 8, 512, hidden=64, seed=9, with_x=True)
ei = b.edge_index.to(dev)
plan = ga.get_plan(ei, n, True)
h = torch.randn(n, 32, generator=torch.Generator().manual_seed(1)).to(dev)
y = ga.spmm(plan, h)
m = ga.GConvAdapter(64, 16, learnable_scalar=True).to(dev)
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for p in m.parameters():
        p.copy_((torch.randn(p.shape, generator=g) * 0.1).to(dev))
x = b.x.to(dev).requires_grad_(True)
out = m(x, ei)
out.backward(torch.randn(out.shape, generator=g).to(dev))
torch.save({{"y": y.cpu(), "out": out.detach().cpu(), "dx": x.grad.cpu(), **{{k: p.grad.cpu() for k, p in m.named_parameters()}}}}, sys.argv[1])
"""


def test_xcd_row_schedule_does_not_change_a_bit(dev, tmp_path):
    """GCONV_SPMM_XCD_ROWS=0 / 1 (read once per process) force the schedule off / on: SpMM output, adapter output
    and every gradient are bit-identical."""
    script = tmp_path / "ab.py"
    script.write_text(_AB_SCRIPT.format(root=ROOT))
    res = {}
    for tag in ("0", "1"):
        out = tmp_path / f"x{tag}.pt"
        subprocess.run([sys.executable, str(script), str(out)], check=True, timeout=600,
                       env={**os.environ, "GCONV_SPMM_XCD_ROWS": tag})
        res[tag] = torch.load(out)
    for k, v in res["0"].items():
        assert torch.equal(res["1"][k], v), f"{k} depends on the row schedule"
