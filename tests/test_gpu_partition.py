"""Row-partitioned adapter (one big graph over several ranks) — `-m gpu`.

Three layers of evidence:
  * every rank's stages run one after the other in ONE process on shared full-height buffers
    (the exchange is then implicit): rows must be BITWISE those of the single-device call,
    whatever the number of ranks, and the parameter gradients must add up;
  * the same against the fp64 oracle (the partition is not only self-consistent);
  * two real processes on one GPU (`gloo`, host-staged exchange) through the nn.Module.
"""
from __future__ import annotations

import ctypes as C
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import gconv_adapter_amd as ga
from gconv_adapter_amd import _lib, partition
from oracle import pyg_restated as oracle

from helpers import assert_close, grad_atol

pytestmark = pytest.mark.gpu


def _graph(kind: str, seed: int):
    if kind == "hubs":        # long rows (segment path) + ordinary rows, symmetric
        b = ga.synth.arxiv_like(num_nodes=6000, num_pairs=24000, hidden=8, seed=seed)
        return b.edge_index, 6000
    if kind == "directed":    # S != S^T: backward runs on the transposed CSR
        g = torch.Generator().manual_seed(seed)
        n = 1500
        return torch.stack([torch.randint(n, (9000,), generator=g), torch.randint(n, (9000,), generator=g)]), n
    if kind == "tiny":        # fewer rows than ranks: empty slabs
        return torch.tensor([[0, 1, 2, 2], [1, 0, 0, 2]]), 3
    raise AssertionError(kind)


def _weights(d, r, seed, dev):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(dev)
    return mk(r, d), mk(r), mk(d, r), mk(d), (1.0 + 0.1 * torch.randn(1, generator=g)).to(dev)


def _single_device(plan, P, x, g, d, r, act, dev):
    lib = _lib.load()
    n = x.shape[0]
    fp, wsf, wsb = plan.sizes(lib, d, r)
    out = torch.empty_like(x)
    saved = torch.zeros(3, n, fp, device=dev)
    wsF = torch.zeros(max(wsf, 16), dtype=torch.uint8, device=dev)
    wsB = torch.zeros(max(wsb, 16), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.gconv_adapter_fwd_f32(plan.ref(), C.byref(P), x.data_ptr(), out.data_ptr(), saved[0].data_ptr(),
                                   saved[2].data_ptr() if act == 2 else None, saved[1].data_ptr(),
                                   wsF.data_ptr(), wsF.numel(), st)
    assert rc == 0, rc
    gx = torch.empty_like(x)
    gp = [torch.empty(r, d, device=dev), torch.empty(r, device=dev), torch.empty(d, r, device=dev),
          torch.empty(d, device=dev), torch.empty(1, device=dev)]
    rc = lib.gconv_adapter_bwd_f32(plan.ref(), C.byref(P), x.data_ptr(), g.data_ptr(), saved[0].data_ptr(),
                                   saved[2].data_ptr() if act == 2 else None, saved[1].data_ptr(), gx.data_ptr(),
                                   *[t.data_ptr() for t in gp], wsB.data_ptr(), wsB.numel(), st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out, saved, gx, gp


def _all_ranks_in_one_process(plan, P, x, g, d, r, act, world, dev):
    """Stage by stage over every rank, on shared buffers (what the exchange would produce)."""
    lib = _lib.load()
    n = x.shape[0]
    fp = int(lib.gconv_padded_rank(r))
    slabs = [ga.RowSlabs(n, world, p) for p in range(world)]
    full = torch.full((6, slabs[0].padded, fp), float("nan"), device=dev)   # h0 h P h1 dp dh1
    h0, sh, sp, sh1, dp, dh1 = full.unbind(0)
    xs, gs_ = [s.take(x) for s in slabs], [s.take(g) for s in slabs]
    outs, gxs = [torch.empty_like(t) for t in xs], [torch.empty_like(t) for t in xs]
    wsF = torch.zeros(max(int(lib.gconv_adapter_slab_fwd_workspace_bytes(plan.ref(), r)), 16),
                      dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for stage in range(3):
        for s, xl, ol in zip(slabs, xs, outs):
            partition.slab_fwd_stage(plan, P, s, stage, xl, ol, h0, sh, sh1 if act == 2 else None, sp, wsF, st)
    grads, wss = [], []
    for s in slabs:
        lo, hi = s.bounds()
        wss.append(torch.zeros(max(int(lib.gconv_adapter_slab_bwd_workspace_bytes(plan.ref(), lo, hi, d, r)), 16),
                               dtype=torch.uint8, device=dev))
        grads.append([torch.full((r, d), float("nan"), device=dev), torch.full((r,), float("nan"), device=dev),
                      torch.full((d, r), float("nan"), device=dev), torch.full((d,), float("nan"), device=dev),
                      torch.full((1,), float("nan"), device=dev)])
    for stage in range(3):
        for s, xl, gl, gxl, gp, ws in zip(slabs, xs, gs_, gxs, grads, wss):
            partition.slab_bwd_stage(plan, P, s, stage, xl, gl, sh, sh1 if act == 2 else None, sp, dp, dh1,
                                     gxl, *gp, ws, st)
    torch.cuda.synchronize()
    return torch.cat(outs), torch.cat(gxs), full[:, :n], grads


@pytest.mark.parametrize("kind,d,r,act,normalize", [
    ("hubs", 128, 32, "relu", True), ("hubs", 64, 16, "silu", True), ("hubs", 32, 5, "relu", False),
    ("directed", 96, 8, "relu", True), ("directed", 40, 12, "none", False), ("tiny", 8, 2, "relu", True)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_row_slabs_reproduce_the_single_device_rows_bitwise(dev, kind, d, r, act, normalize, world):
    ei, n = _graph(kind, seed=d + r)
    plan = ga.GraphPlan(ei.to(dev), n, normalize)
    wd, bd, wu, bu, s = _weights(d, r, seed=r, dev=dev)
    a = _lib.ACT_CODES[act]
    P = _lib.AdapterParams(wd.data_ptr(), bd.data_ptr(), wu.data_ptr(), bu.data_ptr(), s.data_ptr(), d, r, a, 1)
    gen = torch.Generator().manual_seed(7)
    x, g = torch.randn(n, d, generator=gen).to(dev), torch.randn(n, d, generator=gen).to(dev)
    out1, saved1, gx1, gp1 = _single_device(plan, P, x, g, d, r, a, dev)
    out, gx, full, grads = _all_ranks_in_one_process(plan, P, x, g, d, r, a, world, dev)
    if plan.n_tiles == 0:
        assert torch.equal(out, out1) and torch.equal(gx, gx1)
        assert torch.equal(full[1], saved1[0]) and torch.equal(full[2], saved1[1])   # act output, P
        if a == 2:
            assert torch.equal(full[3], saved1[2])
    else:   # the single-device call took the fused tile path (other summation order)
        assert_close(out.cpu().double(), out1.cpu().double(), "out")
        assert_close(gx.cpu().double(), gx1.cpu().double(), "grad x")
    for k, want in enumerate(gp1):   # contributions add up (summation order differs from one device's)
        got = torch.stack([gr[k] for gr in grads]).double().sum(0)
        assert_close(got.cpu(), want.cpu(), f"param grad {k}", 2e-5, grad_atol(want.cpu().double(), 2e-5))
    if world == 1 and plan.n_tiles == 0:   # one slab == the whole graph: the same launches, the same bits
        for k, want in enumerate(gp1):
            assert torch.equal(grads[0][k], want)


def test_row_slabs_against_the_oracle(dev):
    batch = ga.synth.arxiv_like(num_nodes=4000, num_pairs=15000, hidden=64, seed=4)
    n, d, r = 4000, 64, 16
    flags = oracle.AdapterFlags(non_linearity="silu", learnable_scalar=True, normalize=True)
    params = {k: v.float().double() for k, v in oracle.init_params(d, r, flags, seed=2).items()}
    g = torch.randn(n, d, generator=torch.Generator().manual_seed(3))
    want_out, want = oracle.adapter_forward_backward(batch.x.double(), batch.edge_index, params, flags,
                                                     g.double(), need_dx=True)
    f = lambda k: params[k].float().to(dev).contiguous()
    wd, bd, wu, bu, s = f("conv_down.lin.weight"), f("conv_down.bias"), f("conv_up.lin.weight"), \
        f("conv_up.bias"), f("scalar")
    P = _lib.AdapterParams(wd.data_ptr(), bd.data_ptr(), wu.data_ptr(), bu.data_ptr(), s.data_ptr(), d, r, 2, 1)
    plan = ga.GraphPlan(batch.edge_index.to(dev), n, True)
    out, gx, _, grads = _all_ranks_in_one_process(plan, P, batch.x.to(dev), g.to(dev), d, r, 2, 5, dev)
    assert_close(out.cpu().double(), want_out, "out")
    assert_close(gx.cpu().double(), want["x"], "grad x")
    keys = ["conv_down.lin.weight", "conv_down.bias", "conv_up.lin.weight", "conv_up.bias", "scalar"]
    for k, name in enumerate(keys):
        got = torch.stack([gr[k] for gr in grads]).double().sum(0).cpu()
        assert_close(got, want[name], f"grad {name}", 1e-5, grad_atol(want[name]))


def test_slab_entry_points_reject_bad_arguments(dev):
    lib = _lib.load()
    ei, n = _graph("directed", 1)
    plan = ga.GraphPlan(ei.to(dev), n, True)
    wd, bd, wu, bu, s = _weights(16, 4, 0, dev)
    P = _lib.AdapterParams(wd.data_ptr(), bd.data_ptr(), wu.data_ptr(), bu.data_ptr(), s.data_ptr(), 16, 4, 1, 1)
    t = torch.zeros(n, 16, device=dev)
    f = torch.zeros(n, 4, device=dev)
    call = lambda lo, hi, stage: lib.gconv_adapter_slab_fwd_f32(
        plan.ref(), C.byref(P), lo, hi, stage, t.data_ptr(), t.data_ptr(), f.data_ptr(), f.data_ptr(), None,
        f.data_ptr(), None, 0, torch.cuda.current_stream().cuda_stream)
    assert call(0, n + 1, 0) == -1 and call(5, 4, 0) == -1 and call(0, n, 3) == -1 and call(-1, n, 0) == -1
    assert call(7, 7, 1) == 0    # an empty slab is legal and launches nothing
    with pytest.raises(RuntimeError, match="owns"):
        mod = ga.RowSlabGConvAdapter(16, 4).to(dev)
        mod.partition(n, world=2, rank=0)
        mod(t, ei.to(dev))       # x must be this rank's rows, not all of them
    with pytest.raises(RuntimeError, match="partition"):
        ga.RowSlabGConvAdapter(16, 4).to(dev)(t, ei.to(dev))


# ---------------------------------------------------------------------------
# two processes on one GPU through the module (gloo: the exchange is staged through the host)
# ---------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _module_worker(rank, world, port, out_path, reduce_inside):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    batch = ga.synth.arxiv_like(num_nodes=3000, num_pairs=12000, hidden=32, seed=11)
    n, d, r = 3000, 32, 16
    torch.manual_seed(0)
    mod = ga.RowSlabGConvAdapter(d, r, learnable_scalar=True, normalization="layer_norm", normalize=True).to(dev)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, 0.1)
    slabs = mod.partition(n, reduce_param_grads=reduce_inside)
    assert (slabs.world, slabs.rank) == (world, rank)
    x = slabs.take(batch.x.reshape(n, d)).to(dev).requires_grad_(True)
    g = slabs.take(torch.randn(n, d, generator=torch.Generator().manual_seed(1))).to(dev)
    out = mod(x.unsqueeze(0), batch.edge_index.to(dev))      # [1, rows, d], the transductive shape
    out.backward(g.unsqueeze(0))
    if reduce_inside:   # only the parameters outside the fused path are left to the caller
        rest = [p for k, p in mod.named_parameters() if k.startswith("normalization") or k == "scalar"]
        for p in rest:
            dist.all_reduce(p.grad)
    else:               # ONE all-reduce over every trainable parameter
        flat = torch.cat([p.grad.reshape(-1) for p in mod.parameters()])
        dist.all_reduce(flat)
        off = 0
        for p in mod.parameters():
            p.grad = flat[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
    full_out = partition.gather_rows(out[0].detach(), slabs)
    full_gx = partition.gather_rows(x.grad, slabs)
    if rank == 0:
        torch.save({"out": full_out.cpu(), "gx": full_gx.cpu(),
                    "grads": {k: p.grad.cpu() for k, p in mod.named_parameters()}}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("reduce_inside", [False, True])
def test_two_ranks_on_one_gpu_match_the_single_device_module(dev, tmp_path, reduce_inside):
    out_path = str(tmp_path / "r0.pt")
    mp.start_processes(_module_worker, args=(2, _free_port(), out_path, reduce_inside), nprocs=2, join=True,
                       start_method="spawn")
    got = torch.load(out_path)
    batch = ga.synth.arxiv_like(num_nodes=3000, num_pairs=12000, hidden=32, seed=11)
    n, d, r = 3000, 32, 16
    torch.manual_seed(0)
    ref = ga.GConvAdapter(d, r, learnable_scalar=True, normalization="layer_norm", normalize=True).to(dev)
    with torch.no_grad():
        for p in ref.parameters():
            p.normal_(0, 0.1)
    x = batch.x.reshape(n, d).to(dev).requires_grad_(True)
    g = torch.randn(n, d, generator=torch.Generator().manual_seed(1)).to(dev)
    out = ref(x.unsqueeze(0), batch.edge_index.to(dev))
    out.backward(g.unsqueeze(0))
    assert torch.equal(got["out"], out[0].detach().cpu())      # row results do not depend on the partition
    assert_close(got["gx"].double(), x.grad.cpu().double(), "grad x", 1e-5, 1e-6)
    for k, p in ref.named_parameters():
        want = p.grad.cpu().double()
        assert_close(got["grads"][k].double(), want, f"grad {k}", 2e-5, grad_atol(want, 2e-5))


# ---------------------------------------------------------------------------
# molecule batches, sharded BY GRAPH: gradient sinks + ONE asynchronous flat all-reduce
# ---------------------------------------------------------------------------
def _shard_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    batch = ga.synth.molecule_batch("hiv", 48, hidden=64, seed=3)
    shard = ga.synth.shard_graphs(batch, rank, world)
    torch.manual_seed(0)
    mods = torch.nn.ModuleList([ga.GConvAdapter(64, 8, learnable_scalar=True, normalize=True) for _ in range(3)]).to(dev)
    with torch.no_grad():
        for p in mods.parameters():
            p.normal_(0, 0.1)
    params = list(mods.parameters())
    bucket = ga.dist.FlatGradBucket(params)
    assert bucket.sinks, "the C++ binding registers the bucket's views as gradient sinks"
    g_all = torch.randn(batch.num_nodes, 64, generator=torch.Generator().manual_seed(1))
    n0 = int(batch.graph_ptr[(batch.num_graphs * rank) // world])
    g = g_all[n0:n0 + shard.num_nodes].to(dev)
    x, ei = shard.x.to(dev), shard.edge_index.to(dev)
    for step in range(2):                      # the second step overwrites the sinks
        bucket.wait()
        for p in params:
            p.grad = None
        h = x
        for m in mods:
            h = m(h, ei)
        h.backward(g)
        # the backward pass has put every gradient in the bucket: nothing to pack
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, bucket.views))
        assert bucket.gather_stragglers() == 0
        bucket.all_reduce_grads(async_op=True)
    bucket.wait()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({k: p.grad.cpu() for k, p in mods.named_parameters()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(bool(os.environ.get("GCONV_PY_BINDING")) or bool(os.environ.get("GCONV_AUTOGRAD_PARAM_GRADS")),
                    reason="gradient sinks belong to the C++ binding's direct-deposit backward node")
def test_two_ranks_by_graph_with_gradient_sinks(dev, tmp_path):
    """Two processes, one shard of the batch each: the adapters' backward deposits dW /
    db / ds straight into the flat bucket (no cat, no copy back), ONE asynchronous all-reduce per step
    — the summed gradients equal the single-process gradients of the whole batch."""
    out_path = str(tmp_path / "r0.pt")
    mp.start_processes(_shard_worker, args=(2, _free_port(), out_path), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out_path)
    batch = ga.synth.molecule_batch("hiv", 48, hidden=64, seed=3)
    torch.manual_seed(0)
    mods = torch.nn.ModuleList([ga.GConvAdapter(64, 8, learnable_scalar=True, normalize=True) for _ in range(3)]).to(dev)
    with torch.no_grad():
        for p in mods.parameters():
            p.normal_(0, 0.1)
    g = torch.randn(batch.num_nodes, 64, generator=torch.Generator().manual_seed(1)).to(dev)
    h = batch.x.to(dev)
    for m in mods:
        h = m(h, batch.edge_index.to(dev))
    h.backward(g)
    for k, p in mods.named_parameters():
        want = p.grad.cpu().double()
        assert_close(got[k].double(), want, f"grad {k}", 1e-5, grad_atol(want))
