"""BatchNorm over a batch sharded BY GRAPH (SURVEY.md §8e "optional all_reduce of BN stats"): with
`sync_batch_norm` an N-rank step is the single-process step of the whole batch — the reference trains
the backbone's BatchNorm (`/root/reference/src/models/inductive/gnn.py:56,109-110,238-239`) and the
adapter's `batch_norm` tail (`src/finetune/gconv_adapter.py:98-100`) on whole-batch statistics.

Stage tests drive every rank's launches in ONE process through the C ABI (the collective replaced by
`torch.cat` / a sum); the last test runs two real processes on the one GPU (gloo)."""
from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

import gconv_adapter_amd as ga
from gconv_adapter_amd import _lib
from helpers import assert_close, grad_atol

pytestmark = pytest.mark.gpu

D, R, LAYERS, TASKS = 64, 8, 2, 2


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ---------------------------------------------------------------------------
# the four stage entry points, all ranks in one process, against fp64 torch BatchNorm of the whole batch
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("cuts,relu", [((0, 700, 1500), False), ((0, 1, 1499, 1500), True), ((0, 0, 900, 1500, 1500), True),
                                       ((0, 1500), False)])
def test_sharded_batch_norm_stages_match_whole_batch_fp64(dev, cuts, relu):
    lib = _lib.load()
    n, d = cuts[-1], 300
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(n, d, generator=g) * 2.0 + 30.0 * torch.randn(d, generator=g)).float()   # large column offsets
    go = torch.randn(n, d, generator=g).float()
    gamma, beta = torch.randn(d, generator=g).float(), torch.randn(d, generator=g).float()
    rm0, rv0 = torch.randn(d, generator=g).float(), torch.rand(d, generator=g).float() + 0.5
    # fp64 reference on the whole batch
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    yr = F.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    if relu:
        yr = F.relu(yr)
    yr.backward(go.double())

    world = len(cuts) - 1
    words = int(lib.gconv_batch_norm_sync_words(d))
    stream = torch.cuda.current_stream().cuda_stream
    shards, locals_ = [], []
    for r in range(world):
        xs = x[cuts[r]:cuts[r + 1]].contiguous().to(dev)
        ns = xs.shape[0]
        ws = torch.empty(max(int(lib.gconv_batch_norm_workspace_bytes(ns, d)), 16), dtype=torch.uint8, device=dev)
        loc = torch.empty(words, device=dev)
        _lib.check(lib.gconv_batch_norm_local_stats_f32(xs.data_ptr() if ns else None, ns, d, loc.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), stream), "local_stats")
        assert int(loc[2 * d:].view(torch.int32)[0].item()) == ns
        shards.append((xs, ws))
        locals_.append(loc)
    gathered = torch.cat(locals_)                           # what the all-gather hands every rank
    outs, stats_all, sums_loc = [], [], []
    gam, bet = gamma.to(dev), beta.to(dev)
    for r, (xs, ws) in enumerate(shards):
        ns = xs.shape[0]
        y = torch.empty_like(xs)
        stats = torch.empty(2, d, device=dev)
        total = torch.zeros(4, dtype=torch.int32, device=dev)
        rmr, rvr = rm0.to(dev), rv0.to(dev)
        _lib.check(lib.gconv_batch_norm_fwd_sync_f32(xs.data_ptr() if ns else None, gam.data_ptr(), bet.data_ptr(),
                                                     rmr.data_ptr(), rvr.data_ptr(), 0.1, 1e-5, int(relu),
                                                     y.data_ptr() if ns else None, stats[0].data_ptr(),
                                                     stats[1].data_ptr(), total.data_ptr(), ns, d,
                                                     gathered.data_ptr(), world, stream), "fwd_sync")
        assert int(total[0].item()) == n
        if r > 0:                                           # merged in rank order: the same bits everywhere
            assert torch.equal(stats, stats_all[0][0])
        assert_close(rmr.cpu().double(), rm, "running_mean", 1e-5, 1e-5)
        assert_close(rvr.cpu().double(), rv, "running_var", 1e-5, 1e-5)
        gs = go[cuts[r]:cuts[r + 1]].contiguous().to(dev)
        sums = torch.empty(2 * d, device=dev)
        _lib.check(lib.gconv_batch_norm_bwd_local_f32(gs.data_ptr() if ns else None, xs.data_ptr() if ns else None,
                                                      y.data_ptr() if (relu and ns) else None, stats[0].data_ptr(),
                                                      stats[1].data_ptr(), sums.data_ptr(), ns, d, ws.data_ptr(),
                                                      ws.numel(), stream), "bwd_local")
        outs.append(y)
        stats_all.append((stats, total, gs))
        sums_loc.append(sums)
    sums_all = torch.stack(sums_loc).sum(0)                 # what the all-reduce hands every rank
    assert_close(torch.cat(outs).cpu().double(), yr.detach(), "y", 1e-5, 1e-5)
    assert_close(sums_all[:d].cpu().double(), gr.grad, "grad gamma", 1e-5, grad_atol(gr.grad))
    assert_close(sums_all[d:].cpu().double(), br.grad, "grad beta", 1e-5, grad_atol(br.grad))
    dxs = []
    for r, (xs, _) in enumerate(shards):
        ns = xs.shape[0]
        stats, total, gs = stats_all[r]
        dx = torch.empty_like(xs)
        _lib.check(lib.gconv_batch_norm_bwd_dx_f32(gs.data_ptr() if ns else None, xs.data_ptr() if ns else None,
                                                   outs[r].data_ptr() if (relu and ns) else None, stats[0].data_ptr(),
                                                   stats[1].data_ptr(), gam.data_ptr(), sums_all.data_ptr(),
                                                   total.data_ptr(), dx.data_ptr() if ns else None, ns, d, stream),
                   "bwd_dx")
        dxs.append(dx)
    assert_close(torch.cat(dxs).cpu().double(), xr.grad, "grad x", 1e-5, 1e-5)


# ---------------------------------------------------------------------------
# two processes, one shard each: loss and every trainable gradient of the whole-batch step
# ---------------------------------------------------------------------------
class _Model(nn.Module):
    """`MolecularGNN.forward` in miniature (`gnn.py:85-130`): adapter -> frozen GIN -> BatchNorm (-> ReLU);
    the first adapter carries the `batch_norm` tail of ablation 4 (`gconv_adapter.py:98-100`)."""

    def __init__(self):
        super().__init__()
        self.gnn_layers = nn.ModuleList([ga.MolecularGINConv(D) for _ in range(LAYERS)])
        self.batch_norm_layers = nn.ModuleList([ga.BatchNorm1d(D, fused_relu=i != LAYERS - 1) for i in range(LAYERS)])
        self.pre_adapters = nn.ModuleDict({
            str(i): ga.GConvAdapter(D, R, learnable_scalar=True, normalize=True,
                                    normalization="batch_norm" if i == 0 else "none") for i in range(LAYERS)})
        self.graph_pred_linear = nn.Linear(D, TASKS)
        for name, module in self.named_modules():   # freeze rule, gnn.py:230-241
            for _, p in module.named_parameters(recurse=False):
                p.requires_grad = ("adapter" in name or "graph_pred_linear" in name
                                   or isinstance(module, (nn.LayerNorm, nn.BatchNorm1d)))

    def forward(self, x, edge_index, edge_attr, batch, num_graphs):
        h = x
        for layer in range(LAYERS):
            h = self.pre_adapters[str(layer)](h, edge_index, edge_attr)
            h = self.gnn_layers[layer](h, edge_index, edge_attr)
            h = self.batch_norm_layers[layer](h)
        pooled = torch.zeros(num_graphs, D, device=h.device, dtype=h.dtype).index_add_(0, batch, h)
        pooled = pooled / torch.bincount(batch, minlength=num_graphs).clamp(min=1).unsqueeze(1)
        return self.graph_pred_linear(pooled)


def _problem():
    b = ga.synth.molecule_batch("hiv", 40, hidden=D, seed=4)
    g = torch.Generator().manual_seed(3)
    e = b.edge_index.shape[1]
    ea = torch.stack([torch.randint(0, 4, (e,), generator=g), torch.randint(0, 3, (e,), generator=g)], 1)
    y = torch.randint(0, 2, (b.num_graphs, TASKS), generator=g).double()
    y[torch.rand(y.shape, generator=g) < 0.2] = float("nan")
    torch.manual_seed(0)
    model = _Model()
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "adapter" in k and "scalar" not in k and "normalization" not in k:
                p.normal_(0, 0.1)
    return b, ea, y, model


def _step(model, b, ea, y, dev, count):
    sizes = b.graph_ptr[1:] - b.graph_ptr[:-1]
    batch = torch.repeat_interleave(torch.arange(b.num_graphs), sizes).to(dev)
    out = model(b.x.to(dev), b.edge_index.to(dev), ea.to(dev), batch, b.num_graphs)
    y = y.to(dev)
    valid = ~torch.isnan(y)
    loss = F.binary_cross_entropy_with_logits(out[valid].double(), y[valid], reduction="none").sum() / count
    loss.backward()
    return loss.detach(), out.detach()


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    b, ea, y, model = _problem()
    model = ga.sync_batch_norm(model.to(dev)).train()
    assert all(m.process_group is not None for m in model.modules() if isinstance(m, ga.BatchNorm1d))
    shard = ga.synth.shard_graphs(b, rank, world)
    lo, hi = (b.num_graphs * rank) // world, (b.num_graphs * (rank + 1)) // world
    n0, n1 = int(b.graph_ptr[lo]), int(b.graph_ptr[hi])
    keep = (b.edge_index[1] >= n0) & (b.edge_index[1] < n1)
    ys = y[lo:hi]
    # the loss is divided by the valid-target count of the WHOLE batch (finetune_inductive_learning.py:63)
    count = ga.dist.global_count((~torch.isnan(ys)).sum().to(dev))
    params = ga.dist.adapter_trainable_parameters(model)
    bucket = ga.dist.FlatGradBucket(params)
    for p in params:
        p.grad = None
    loss, out = _step(model, shard, ea[keep], ys, dev, count)
    bucket.all_reduce_grads()
    loss = loss.clone()
    dist.all_reduce(loss_cpu := loss.cpu())
    outs = [None] * world
    dist.all_gather_object(outs, out.cpu())
    if rank == 0:
        torch.save({"loss": loss_cpu, "out": torch.cat(outs),
                    "grads": {k: p.grad.cpu() for k, p in model.named_parameters() if p.requires_grad},
                    "buffers": {k: v.cpu() for k, v in model.named_buffers()}}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_sync_batch_norm_equal_the_whole_batch_step(dev, tmp_path):
    out_path = str(tmp_path / "r0.pt")
    mp.start_processes(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out_path)
    b, ea, y, model = _problem()
    model = model.to(dev).train()
    loss, out = _step(model, b, ea, y, dev, (~torch.isnan(y)).sum().to(dev))
    assert abs(got["loss"].item() - loss.item()) <= 1e-5 * max(1.0, abs(loss.item()))
    assert_close(got["out"].double(), out.cpu().double(), "model output", 1e-5, 1e-5)
    for k, v in model.named_buffers():   # running statistics of the whole batch on every rank
        if v.dtype.is_floating_point:
            assert_close(got["buffers"][k].double(), v.cpu().double(), k, 1e-5, 1e-5)
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        want = p.grad.cpu().double()
        assert_close(got["grads"][k].double(), want, f"grad {k}", 1e-5, grad_atol(want))


# ---------------------------------------------------------------------------
# the whole loop body over several steps: sync BatchNorm + FusedAdam whose flat gradient buffer is the
# all-reduce bucket — the parameters after k steps equal the single-process run's
# ---------------------------------------------------------------------------
STEPS = 3


def _train(model, b, ea, y, dev, count_fn, reduce):
    # eps = 1e-3: Adam's g / (sqrt(v) + eps) otherwise turns a gradient entry of ~1e-7 — pure rounding noise, of
    # either sign — into a full-size step of +-lr, and the two runs agree on gradients to 1e-5, not on noise
    opt = ga.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-2, eps=1e-3)
    for _ in range(STEPS):
        opt.zero_grad(set_to_none=True)
        _step(model, b, ea, y, dev, count_fn())
        if reduce:
            opt.all_reduce_grads(async_op=True)      # step() waits for it
        opt.step()
    return {k: p.detach().cpu() for k, p in model.named_parameters() if p.requires_grad}


def _train_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    b, ea, y, model = _problem()
    model = ga.sync_batch_norm(model.to(dev)).train()
    shard = ga.synth.shard_graphs(b, rank, world)
    lo, hi = (b.num_graphs * rank) // world, (b.num_graphs * (rank + 1)) // world
    n0, n1 = int(b.graph_ptr[lo]), int(b.graph_ptr[hi])
    keep = (b.edge_index[1] >= n0) & (b.edge_index[1] < n1)
    ys = y[lo:hi]
    got = _train(model, shard, ea[keep], ys, dev,
                 lambda: ga.dist.global_count((~torch.isnan(ys)).sum().to(dev)), reduce=True)
    if rank == 0:
        torch.save(got, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_train_like_one_process(dev, tmp_path):
    out_path = str(tmp_path / "r0.pt")
    mp.start_processes(_train_worker, args=(2, _free_port(), out_path), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out_path)
    b, ea, y, model = _problem()
    want = _train(model.to(dev).train(), b, ea, y, dev, lambda: (~torch.isnan(y)).sum().to(dev), reduce=False)
    for k, w in want.items():
        assert_close(got[k].double(), w.double(), f"param {k} after {STEPS} steps", 1e-5, 1e-5 * max(1.0, float(w.abs().max())))


def test_without_a_process_group_the_layer_is_local(dev):
    m = ga.BatchNorm1d(32).to(dev)
    assert m.process_group is None
    with pytest.raises(RuntimeError, match="process group"):
        ga.sync_batch_norm(m)
    import copy
    m.process_group = object()
    assert copy.deepcopy(m).process_group is None       # a group handle does not travel with a copy
