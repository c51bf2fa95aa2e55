"""The evaluation path: the reference runs the model forward-only, in eval mode and under
``torch.no_grad()``, three times per epoch (``/root/reference/scripts/finetune_inductive_learning.py:78-152,
401-409``: ``model.eval()`` ``:96``, ``with torch.no_grad()`` ``:103-104``). Here that call takes
``gconv_adapter_infer_f32`` (no saved tensors) and, for ``normalization='batch_norm'``, the running statistics
through the folded-scalar BatchNorm tail (``adapter.py``). Checked against the fp64 oracle, whose ``training``
argument follows ``nn.BatchNorm1d`` (``oracle/pyg_restated.py:adapter_forward``)."""
from __future__ import annotations

import os
import subprocess
import sys

import pytest
import torch

import gconv_adapter_amd as ga
from oracle import pyg_restated as oracle

from helpers import assert_close, build_module

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(d, r, flags, seed):
    return {k: v.float().double() for k, v in oracle.init_params(d, r, flags, seed=seed).items()}


@pytest.mark.parametrize("scalar", [True, False])
@pytest.mark.parametrize("normalization", ["none", "layer_norm", "batch_norm"])
@pytest.mark.parametrize("act", ["relu", "silu"])
def test_eval_forward_matches_the_oracle(dev, normalization, scalar, act):
    """`.eval()` + `no_grad()` on a molecule batch (fused tile path). batch_norm: two TRAINING steps on two other
    batches first — module and oracle both update their running statistics (momentum 0.1, unbiased variance) —
    then the evaluation batch is normalised with those statistics on both sides."""
    d, r = 300, 8
    flags = oracle.AdapterFlags(non_linearity=act, normalization=normalization, learnable_scalar=scalar)
    params = _params(d, r, flags, seed=3)
    mod = build_module(ga, d, r, flags, params, dev)
    if normalization == "batch_norm":
        params["normalization.running_mean"] = torch.zeros(d, dtype=torch.float64)
        params["normalization.running_var"] = torch.ones(d, dtype=torch.float64)
        for step in range(2):
            tb = ga.synth.molecule_batch("esol", 24 + 8 * step, hidden=d, seed=10 + step)
            oracle.adapter_forward(tb.x.double() + 0.3 * step, tb.edge_index, params, flags, training=True)
            mod(tb.x.to(dev) + 0.3 * step, tb.edge_index.to(dev))
        assert_close(mod.normalization.running_mean.cpu().double(), params["normalization.running_mean"], "running_mean")
        assert_close(mod.normalization.running_var.cpu().double(), params["normalization.running_var"], "running_var")
        assert int(mod.normalization.num_batches_tracked) == 2
    batch = ga.synth.molecule_batch("esol", 32, hidden=d, seed=5)
    want = oracle.adapter_forward(batch.x.double(), batch.edge_index, params, flags, training=False)
    mod.eval()
    with torch.no_grad():
        got = mod(batch.x.to(dev), batch.edge_index.to(dev))
    assert not got.requires_grad and got.grad_fn is None
    assert_close(got.cpu().double(), want, "eval out")
    if normalization == "batch_norm":   # evaluation leaves the statistics alone
        assert int(mod.normalization.num_batches_tracked) == 2
        assert_close(mod.normalization.running_mean.cpu().double(), params["normalization.running_mean"], "running_mean")


@pytest.mark.parametrize("kind", ["esol", "hiv128", "cora3d", "hubs_silu", "directed"])
def test_forward_only_call_gives_the_training_forward_bit_for_bit(dev, kind):
    """`gconv_adapter_infer_f32` runs the same launches as `gconv_adapter_fwd_f32`: `out` under `no_grad()` equals
    the output of a recorded forward bit for bit — tile path, gather-fused and four-launch general paths, a
    `[1, N, d]` input — and the recorded forward still trains afterwards (the plan is shared)."""
    act, normalize = "relu", True
    if kind == "esol":
        b, cfg = ga.synth.make_workload("esol"); d, r = cfg["d"], cfg["r"]; x, ei = b.x, b.edge_index
    elif kind == "hiv128":
        b, cfg = ga.synth.make_workload("hiv", graphs=128); d, r = cfg["d"], cfg["r"]; x, ei = b.x, b.edge_index
    elif kind == "cora3d":
        b, cfg = ga.synth.make_workload("cora"); d, r = cfg["d"], cfg["r"]; x, ei = b.x, b.edge_index   # [1, N, 32]
    elif kind == "hubs_silu":
        b = ga.synth.arxiv_like(6000, 30000, hidden=64, seed=2); d, r, act = 64, 16, "silu"; x, ei = b.x, b.edge_index
    else:
        g = torch.Generator().manual_seed(4)
        n, d, r, normalize = 900, 128, 5, False
        ei = torch.stack([torch.randint(n, (5000,), generator=g), torch.randint(n, (5000,), generator=g)])
        x = torch.randn(n, d, generator=g)
    flags = oracle.AdapterFlags(non_linearity=act, learnable_scalar=True, normalize=normalize)
    mod = build_module(ga, d, r, flags, _params(d, r, flags, seed=1), dev)
    xd, eid = x.to(dev), ei.to(dev)
    rec = mod(xd, eid)
    assert rec.requires_grad
    with torch.no_grad():
        inf = mod(xd, eid)
    assert not inf.requires_grad and inf.shape == rec.shape
    assert torch.equal(inf, rec.detach())
    rec.square().mean().backward()                      # the recorded forward's saved tensors are intact
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mod.parameters())
    for p in mod.parameters():                          # a frozen module with grad mode ON: forward-only as well
        p.requires_grad_(False)
    frozen = mod(xd, eid)
    assert not frozen.requires_grad and torch.equal(frozen, inf)
    xg = xd.clone().requires_grad_(True)                # ... unless x itself needs a gradient
    assert mod(xg, eid).requires_grad


def test_forward_only_through_the_ctypes_binding_and_the_c_abi(dev):
    """The same entry point through the ctypes binding (`functional.adapter_infer`), its workspace contract and its
    argument checks."""
    import ctypes as C
    b, cfg = ga.synth.make_workload("pubmed")
    n, d, r = b.num_nodes, cfg["d"], cfg["r"]
    flags = oracle.AdapterFlags(learnable_scalar=True)
    mod = build_module(ga, d, r, flags, _params(d, r, flags, seed=2), dev)
    x, ei = b.x.reshape(n, d).to(dev), b.edge_index.to(dev)
    plan = ga.get_plan(ei, n, True)
    lib = ga._lib.load()
    with torch.no_grad():
        want = mod(x, ei)
    got = ga.functional.adapter_infer(x, mod.conv_down.lin.weight, mod.conv_down.bias, mod.conv_up.lin.weight,
                                      mod.conv_up.bias, mod.scalar, plan, 1, True)
    assert torch.equal(got, want)
    need = int(lib.gconv_adapter_infer_workspace_bytes(plan.ref(), d, r))
    assert need >= int(lib.gconv_adapter_fwd_workspace_bytes(plan.ref(), d, r)) + 2 * n * 16 * 4   # general path: H2', P
    P = ga._lib.AdapterParams(mod.conv_down.lin.weight.data_ptr(), mod.conv_down.bias.data_ptr(),
                              mod.conv_up.lin.weight.data_ptr(), mod.conv_up.bias.data_ptr(), mod.scalar.data_ptr(),
                              d, r, 1, 1)
    out = torch.empty_like(x)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.gconv_adapter_infer_f32(plan.ref(), C.byref(P), x.data_ptr(), out.data_ptr(), ws.data_ptr(), need // 2, st) == -3
    assert lib.gconv_adapter_infer_f32(plan.ref(), C.byref(P), x.data_ptr(), None, ws.data_ptr(), need, st) == -1
    assert lib.gconv_adapter_infer_f32(plan.ref(), C.byref(P), x.data_ptr(), out.data_ptr(), ws.data_ptr(), need, st) == 0
    assert torch.equal(out, want)
    if os.environ.get("GCONV_NO_TILE_PATH"):
        return
    # a molecule batch keeps its intermediates in LDS: no workspace beyond the forward's placeholder
    mb, mcfg = ga.synth.make_workload("esol")
    mplan = ga.get_plan(mb.edge_index.to(dev), mb.num_nodes, True)
    assert int(lib.gconv_adapter_infer_workspace_bytes(mplan.ref(), mcfg["d"], mcfg["r"])) == \
        int(lib.gconv_adapter_fwd_workspace_bytes(mplan.ref(), mcfg["d"], mcfg["r"]))


@pytest.mark.skipif(bool(os.environ.get("GCONV_NO_TILE_PATH")), reason="about the fused tile kernels")
def test_tile_walk_order_does_not_change_the_outputs(dev):
    """Round 5: with two or more tiles per persistent workgroup HALF the workgroups walk their tiles smallest first,
    and `gconv_adapter_params_t::walk` picks the half (the forward's; the backward takes the other): consecutive
    adapters alternate (`GraphPlan.next_walk`), so a workgroup starts a kernel with the tile it ended the previous
    kernel with. A tile's rows have one writer and its arithmetic does not depend on when it runs: `out` and dX are
    bit-identical for walk = 0 / 1; the parameter gradients are sums of per-workgroup partials whose tiles are then
    added in another order — equal at the parameter-gradient tolerance, and bit-identical for the SAME walk."""
    from helpers import grad_atol
    b = ga.synth.molecule_batch("hiv", 2700, hidden=300, seed=7)
    ei = b.edge_index.to(dev)
    plan = ga.get_plan(ei, b.num_nodes, True)
    assert plan.n_tiles >= 512                                  # at least two passes: the order matters at all
    mod = ga.GConvAdapter(300, 8, learnable_scalar=True).to(dev)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_((torch.randn(p.shape, generator=g) * 0.1).to(dev))
    gout = torch.randn(b.num_nodes, 300, generator=g).to(dev)

    def run(walk):
        mod.walk = walk
        x = b.x.to(dev).requires_grad_(True)
        mod.zero_grad(set_to_none=True)
        out = mod(x, ei)
        out.backward(gout)
        return {"out": out.detach(), "dx": x.grad, **{k: p.grad.clone() for k, p in mod.named_parameters()}}
    r0, r1, r0b = run(0), run(1), run(0)
    for k in r0:
        assert torch.equal(r0[k], r0b[k]), f"{k}: not reproducible for the same walk"
    assert torch.equal(r0["out"], r1["out"]) and torch.equal(r0["dx"], r1["dx"])
    for k, v in r0.items():
        assert_close(r1[k].double().cpu(), v.double().cpu(), k, atol=grad_atol(v.double().cpu()))
    # automatic mode: the plan hands out 0, 1, 0, ... in call order
    mod.walk = None
    w0 = plan._walk
    with torch.no_grad():
        mod(b.x.to(dev), ei); mod(b.x.to(dev), ei)
    assert plan._walk == w0 and plan.next_walk() == w0 and plan.next_walk() == w0 ^ 1


@pytest.mark.parametrize("kind", ["tile", "general"])
def test_completion_event_of_a_forward_orders_a_prefetched_build(dev, kind):
    """Round 5 (ABI 11): `gconv_arm_completion_event` — the next forward of the thread records the event at the
    completion of its last kernel (tile path: the kernel's own completion signal; general path: a record behind
    the last launch), and `prefetch_plan(..., ready=event)` makes the build's side stream wait for it without
    queueing anything on the caller's stream. The plan built that way equals the foreground plan."""
    if ga._torch_ext.get() is None:
        pytest.skip("needs the native binding")
    if kind == "tile":
        b = ga.synth.molecule_batch("hiv", 600, hidden=300, seed=3); d, r = 300, 8
        nxt = ga.synth.molecule_batch("hiv", 500, hidden=8, seed=4)
    else:
        b = ga.synth.arxiv_like(20_000, 80_000, hidden=64, seed=1); d, r = 64, 16
        nxt = ga.synth.arxiv_like(9_000, 30_000, hidden=8, seed=2)
    mod = ga.GConvAdapter(d, r, learnable_scalar=True).to(dev)
    x, ei = b.x.to(dev), b.edge_index.to(dev)
    nei = nxt.edge_index.to(dev)
    plan = ga.get_plan(ei, b.num_nodes, True)
    assert (plan.n_tiles > 0) == (kind == "tile")       # (GCONV_NO_TILE_PATH: the same plan, the general path's record)
    torch.cuda.synchronize()
    ev = ga.CompletionEvent()
    assert ev.done()                       # never armed: nothing to wait for
    want = mod(x, ei)
    for _ in range(3):                     # re-armed and re-recorded, as a training loop does
        ev.arm()
        out = mod(x, ei)
        ga.prefetch_plan(nei, nxt.num_nodes, True, ready=ev)
        out.square().mean().backward()
        got = ga.get_plan(nei, nxt.num_nodes, True)
        ref = ga.GraphPlan(nei, nxt.num_nodes, True)
        assert torch.equal(out, want)
        assert got.n_tiles == ref.n_tiles and got.fwd.nnz == ref.fwd.nnz
        h = torch.randn(nxt.num_nodes, 8, generator=torch.Generator().manual_seed(1)).to(dev)
        assert torch.equal(ga.spmm(got, h), ga.spmm(ref, h))
        ga.PLAN_CACHE.discard(nei, nxt.num_nodes, True)
    torch.cuda.synchronize()
    assert ev.done()
    ev.arm()                               # armed, then a call that is rejected: the next good call still works
    with pytest.raises(RuntimeError):
        mod(x[:-1], ei)
    y = mod(x, ei)
    torch.cuda.synchronize()
    assert torch.equal(y, want) and ev.done()
