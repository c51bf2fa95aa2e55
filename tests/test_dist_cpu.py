"""world_size-2 `gloo` tests of the multi-GPU host logic on CPU (`-m "not gpu"`):
graph sharding + ONE flat all-reduce reproduce the single-process gradient."""
from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gconv_adapter_amd as ga


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _ToyAdapter(torch.nn.Module):
    """Per-node stand-in with the adapter's parameter layout: the point of this test is
    the bucket / sharding / loss-normalisation plumbing, which is device-agnostic (the
    HIP compute path itself is covered by the gpu tests)."""

    def __init__(self, d, r):
        super().__init__()
        self.a = ga.GConvAdapter(d, r, learnable_scalar=True)
        with torch.no_grad():
            for p in self.a.parameters():
                p.normal_(0, 0.1)

    def forward(self, x):
        a = self.a
        h = torch.relu(x @ a.conv_down.lin.weight.T + a.conv_down.bias)
        return (h @ a.conv_up.lin.weight.T + a.conv_up.bias + x) * a.scalar


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    assert ga.dist.init_process_group_from_env("gloo") == world
    torch.manual_seed(0)
    model = _ToyAdapter(16, 4)
    batch = ga.synth.molecule_batch("esol", 8, hidden=16, seed=5)
    shard = ga.synth.shard_graphs(batch, rank, world)
    bucket = ga.dist.FlatGradBucket(ga.dist.adapter_trainable_parameters(model))
    views = [p.grad.data_ptr() for p in bucket.params]
    bucket.zero_()
    target = torch.ones(shard.num_nodes, 16)
    count = ga.dist.global_count(torch.tensor(float(shard.num_nodes * 16)))
    loss = ((model(shard.x) - target) ** 2).sum() / count       # divided by the GLOBAL count
    loss.backward()
    assert [p.grad.data_ptr() for p in bucket.params] == views, "autograd must accumulate in place"
    bucket.all_reduce()
    flat_views = torch.cat([v.flatten() for v in bucket.views]).clone()
    # steps that start from p.grad = None: autograd's fresh tensors are packed with one multi-tensor
    # copy, ONE all-reduce (asynchronous here), p.grad = views of the flat buffer (nothing copied back)
    for p in bucket.params:
        p.grad = None
    loss = ((model(shard.x) - target) ** 2).sum() / count
    loss.backward()
    assert bucket.all_reduce_grads(async_op=True) is not None
    bucket.wait()
    assert [p.grad.data_ptr() for p in bucket.params] == views
    packed = torch.cat([p.grad.flatten() for p in bucket.params])
    assert torch.allclose(packed, flat_views, rtol=1e-6, atol=1e-7)
    assert bucket.gather_stragglers() == 0
    # average=True with async_op=True: the division follows the sum, in wait() (round-3 advisor finding:
    # it was silently dropped); and the plain all_reduce() packs gradients autograd allocated itself
    # (a bucket in sink mode on a device without the binding, head / norm parameters) before it reduces
    for p in bucket.params:
        p.grad = None
    loss = ((model(shard.x) - target) ** 2).sum() / count
    loss.backward()
    assert bucket.all_reduce(average=True, async_op=True) is not None
    bucket.wait()
    assert [p.grad.data_ptr() for p in bucket.params] == views
    mean = torch.cat([p.grad.flatten() for p in bucket.params])
    assert torch.allclose(mean * world, flat_views, rtol=1e-6, atol=1e-7)
    if rank == 0:
        torch.save({"flat": flat_views, "count": count}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_flat_allreduce_matches_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.start_processes(_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out)
    torch.manual_seed(0)
    model = _ToyAdapter(16, 4)
    batch = ga.synth.molecule_batch("esol", 8, hidden=16, seed=5)
    loss = ((model(batch.x) - torch.ones(batch.num_nodes, 16)) ** 2).sum() / (batch.num_nodes * 16)
    loss.backward()
    want = torch.cat([p.grad.flatten() for p in ga.dist.adapter_trainable_parameters(model)])
    assert got["count"].item() == batch.num_nodes * 16
    assert torch.allclose(got["flat"], want, rtol=1e-5, atol=1e-6)


def test_bucket_layout_single_process():
    m = ga.GConvAdapter(8, 2, learnable_scalar=True)
    b = ga.dist.FlatGradBucket(m.parameters())
    # every parameter on a 16-byte boundary (the adapter kernels read / write them 4 floats at a time)
    assert b.flat.numel() == sum(-(-p.numel() // 4) * 4 for p in m.parameters()) and b.nbytes == 4 * b.flat.numel()
    assert not b.sinks                       # CPU tensors: bind() mode
    for p, off in zip(b.params, b.offsets):
        assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * off and off % 4 == 0
    b.flat.fill_(3.0)
    b.zero_()
    assert all(torch.count_nonzero(p.grad) == 0 for p in m.parameters())
    assert b.all_reduce() is None            # no process group: no-op
    assert ga.dist.shard_bounds(10, 1, 4) == (2, 5)
    with pytest.raises(ValueError):
        ga.dist.FlatGradBucket([])


# ---------------------------------------------------------------------------
# row partition of one big graph: slab arithmetic and the exchange, on CPU
# ---------------------------------------------------------------------------
def test_row_slabs_cover_every_row_once():
    for n, world in ((10, 4), (170_000, 8), (3, 8), (0, 2), (64, 1), (7, 7)):
        slabs = [ga.RowSlabs(n, world, r) for r in range(world)]
        assert all(s.padded == world * s.slab and s.padded >= n for s in slabs)
        covered = [i for s in slabs for i in range(*s.bounds())]
        assert covered == list(range(n))                       # contiguous, disjoint, in rank order
        assert all(s.lo == min(s.rank * s.slab, n) for s in slabs)   # global row == buffer row
        assert all(s.rows <= s.slab for s in slabs)
    with pytest.raises(ValueError):
        ga.RowSlabs(5, 2, 2)
    t = torch.arange(20.).view(10, 2)
    assert torch.equal(ga.RowSlabs(10, 4, 3).take(t), t[9:10])
    full = torch.zeros(12, 2)
    ga.partition.exchange_rows(full, ga.RowSlabs(12, 1, 0))    # world 1: nothing to exchange
    with pytest.raises(RuntimeError, match="process group"):
        ga.partition.exchange_rows(full, ga.RowSlabs(12, 2, 0))


def _exchange_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    assert ga.dist.init_process_group_from_env("gloo") == world
    n, f = 11, 4                                       # last slab is short: 6 + 5 rows, padded to 12
    slabs = ga.RowSlabs(n, world, rank)
    want = torch.arange(float(n * f)).view(n, f)
    full = torch.full((slabs.padded, f), -1.0)
    lo, hi = slabs.bounds()
    full[lo:hi] = want[lo:hi]                          # a stage has written this rank's rows
    ga.partition.exchange_rows(full, slabs)
    assert torch.equal(full[:n], want)
    got = ga.partition.gather_rows(slabs.take(want), slabs)
    assert torch.equal(got, want)
    with pytest.raises(RuntimeError, match="contiguous"):
        ga.partition.exchange_rows(torch.zeros(n, f), slabs)   # not the padded height
    if rank == 0:
        torch.save(full, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_rows_two_ranks_gloo(tmp_path):
    out = str(tmp_path / "full.pt")
    mp.start_processes(_exchange_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    full = torch.load(out)
    assert torch.equal(full[:11], torch.arange(44.).view(11, 4))
