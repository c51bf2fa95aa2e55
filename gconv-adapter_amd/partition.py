"""One big graph over several GPUs: 1-D row partition of the adapter.

The reference trains on ONE device (``scripts/finetune_transductive_learning.py:208-230``
runs ``GConvAdapter.forward``, ``src/finetune/gconv_adapter.py:80-108``, on the whole
citation graph); this module is new. A single graph does not shard by component the way a
molecule batch does, so rank p owns the rows ``[p * slab, (p + 1) * slab)`` of X / out / G /
dX, every rank holds the whole graph plan (index data only), and the two propagations of each
pass read the rank-width operand of ALL rows. That operand is what gets exchanged:

    forward    h0       [N, Fp]   all-gather      backward   dP'   [N, Fp]   all-gather
               act(..)  [N, Fp]   all-gather                 dH1'  [N, Fp]   all-gather
                                                             parameter gradients: all-reduce

Four all-gathers of N * Fp floats per adapter and step (S-arxiv, Fp = 32: 21.8 MB each,
2.7 MB per rank at 8 GPUs) plus one all-reduce of 2rd + r + d + 1 floats: the low rank is what
makes the partition cheap; the d-wide arrays never leave their device.

The kernels are the single-device ones run on a row range (``gconv_adapter_slab_*`` in
``include/gconv_adapter.h``); the library never calls a collective, this module does, between
the stages: RCCL ``all_gather_into_tensor`` IN PLACE on the full-height buffer (backend
"nccl"), or a host-staged ``all_gather`` under "gloo" (CPU rehearsals and the tests, which run
several ranks on one GPU).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib
from .adapter import GConvAdapter
from .functional import _ptr, _require_hip_f32, _stream
from .graph import GraphPlan, get_plan


class RowSlabs:
    """Equal-height row slabs of an N-node graph over ``world`` ranks (the last ones may be
    short or empty); ``padded = world * slab`` is the height of the exchanged buffers."""

    def __init__(self, num_nodes: int, world: int, rank: int):
        if world < 1 or not 0 <= rank < world or num_nodes < 0:
            raise ValueError(f"bad partition: num_nodes={num_nodes} world={world} rank={rank}")
        self.num_nodes, self.world, self.rank = int(num_nodes), int(world), int(rank)
        self.slab = max(1, -(-self.num_nodes // self.world))
        self.padded = self.slab * self.world

    def bounds(self, rank: Optional[int] = None):
        r = self.rank if rank is None else rank
        lo = min(r * self.slab, self.num_nodes)
        return lo, min(lo + self.slab, self.num_nodes)

    @property
    def lo(self) -> int:
        return self.bounds()[0]

    @property
    def hi(self) -> int:
        return self.bounds()[1]

    @property
    def rows(self) -> int:
        lo, hi = self.bounds()
        return hi - lo

    def for_rank(self, rank: int) -> "RowSlabs":
        return RowSlabs(self.num_nodes, self.world, rank)

    def take(self, full: torch.Tensor) -> torch.Tensor:
        """This rank's rows of a full-height ``[N, ...]`` tensor (contiguous copy)."""
        lo, hi = self.bounds()
        return full[lo:hi].contiguous()

    def __repr__(self) -> str:
        return f"RowSlabs(N={self.num_nodes}, world={self.world}, rank={self.rank}, rows=[{self.lo}, {self.hi}))"


def _active(group) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def exchange_rows(full: torch.Tensor, slabs: RowSlabs, group=None) -> None:
    """All-gather of a ``[slabs.padded, F]`` buffer in place: every rank contributes the slab it
    has just written. One RCCL call; nothing is packed or copied on the nccl backend."""
    if slabs.world == 1:
        return
    if not _active(group):
        raise RuntimeError("exchange_rows: world > 1 needs an initialised process group")
    if full.shape[0] != slabs.padded or not full.is_contiguous():
        raise RuntimeError(f"exchange buffer must be contiguous [{slabs.padded}, F], got {tuple(full.shape)}")
    mine = full[slabs.rank * slabs.slab:(slabs.rank + 1) * slabs.slab]
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(full, mine, group=group)
        return
    host = mine.cpu()   # gloo has no device all-gather: rehearsal / test path
    parts = [torch.empty_like(host) for _ in range(slabs.world)]
    dist.all_gather(parts, host, group=group)
    full.copy_(torch.cat(parts), non_blocking=False)


def gather_rows(local: torch.Tensor, slabs: RowSlabs, group=None) -> torch.Tensor:
    """Full-height ``[N, ...]`` tensor from every rank's rows (evaluation / tests)."""
    buf = torch.zeros((slabs.padded,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    lo, hi = slabs.bounds()
    buf[slabs.rank * slabs.slab:slabs.rank * slabs.slab + (hi - lo)] = local
    flat = buf.view(slabs.padded, -1)
    exchange_rows(flat, slabs, group)
    return buf[:slabs.num_nodes]


# ---------------------------------------------------------------------------
# the stages (thin wrappers of the C ABI; also driven directly by the parity tests)
# ---------------------------------------------------------------------------
def slab_fwd_stage(plan: GraphPlan, params: "_lib.AdapterParams", slabs: RowSlabs, stage: int,
                   x_loc, out_loc, h0, saved_h, saved_h1, saved_p, ws, stream: int) -> None:
    lo, hi = slabs.bounds()
    st = _lib.load().gconv_adapter_slab_fwd_f32(
        plan.ref(), C.byref(params), lo, hi, stage, x_loc.data_ptr(), out_loc.data_ptr(),
        h0.data_ptr(), saved_h.data_ptr(), _ptr(saved_h1), saved_p.data_ptr(),
        ws.data_ptr(), ws.numel(), stream)
    if st != 0:
        _lib.check(st, f"adapter_slab_fwd(stage {stage})")


def slab_bwd_stage(plan: GraphPlan, params: "_lib.AdapterParams", slabs: RowSlabs, stage: int,
                   x_loc, g_loc, saved_h, saved_h1, saved_p, dp, dh1, gx, gwd, gbd, gwu, gbu, gs,
                   ws, stream: int) -> None:
    lo, hi = slabs.bounds()
    st = _lib.load().gconv_adapter_slab_bwd_f32(
        plan.ref(), C.byref(params), lo, hi, stage, x_loc.data_ptr(), g_loc.data_ptr(),
        saved_h.data_ptr(), _ptr(saved_h1), saved_p.data_ptr(), dp.data_ptr(), dh1.data_ptr(),
        _ptr(gx), gwd.data_ptr(), _ptr(gbd), gwu.data_ptr(), _ptr(gbu), _ptr(gs),
        ws.data_ptr(), ws.numel(), stream)
    if st != 0:
        _lib.check(st, f"adapter_slab_bwd(stage {stage})")


def _flat_param_grads(w_down, b_down, w_up, b_up, scalar):
    """The five gradient tensors as views of ONE flat buffer (16-byte aligned pieces), so that
    the cross-rank sum is a single all-reduce."""
    shapes = [w_down.shape, None if b_down is None else b_down.shape, w_up.shape,
              None if b_up is None else b_up.shape, None if scalar is None else scalar.shape]
    offs, total = [], 0
    for s in shapes:
        offs.append(total)
        if s is not None:
            total += -(-int(torch.Size(s).numel()) // 4) * 4
    flat = torch.empty(total, dtype=torch.float32, device=w_down.device)
    views = [None if s is None else flat[o:o + torch.Size(s).numel()].view(s) for s, o in zip(shapes, offs)]
    return flat, views


class RowSlabAdapterFunction(torch.autograd.Function):
    """The adapter on this rank's rows; exchanges between the stages (module docstring)."""

    @staticmethod
    def forward(ctx, x, w_down, b_down, w_up, b_up, scalar, plan: GraphPlan, slabs: RowSlabs,
                act: int, skip: bool, group, reduce_grads: bool):
        lib = _lib.load()
        _require_hip_f32(x, "x")
        x = x.contiguous()
        w_down, w_up = w_down.contiguous(), w_up.contiguous()
        n_loc, d = x.shape
        r = w_down.shape[0]
        if plan.num_nodes != slabs.num_nodes:
            raise RuntimeError(f"graph plan has {plan.num_nodes} nodes, the partition {slabs.num_nodes}")
        if n_loc != slabs.rows:
            raise RuntimeError(f"x has {n_loc} rows but this rank owns {slabs.rows} ({slabs})")
        fp = int(lib.gconv_padded_rank(r))
        if fp == 0:
            raise RuntimeError(f"bottleneck_size={r} is unsupported (1..64)")
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        out = torch.empty_like(x)
        # full-height, global row coordinates: h0, act output, P (+ pre-activation for SiLU)
        full = torch.empty(4 if act == 2 else 3, slabs.padded, fp, dtype=torch.float32, device=dev)
        h0, saved_h, saved_p = full[0], full[1], full[2]
        saved_h1 = full[3] if act == 2 else None
        ws = torch.empty(max(int(lib.gconv_adapter_slab_fwd_workspace_bytes(plan.ref(), r)), 16),
                         dtype=torch.uint8, device=dev)
        params = _lib.AdapterParams(w_down.data_ptr(), _ptr(b_down), w_up.data_ptr(), _ptr(b_up),
                                    _ptr(scalar), d, r, act, int(skip))
        stream = _stream(dev.index)
        slab_fwd_stage(plan, params, slabs, 0, x, out, h0, saved_h, saved_h1, saved_p, ws, stream)
        exchange_rows(h0, slabs, group)
        slab_fwd_stage(plan, params, slabs, 1, x, out, h0, saved_h, saved_h1, saved_p, ws, stream)
        exchange_rows(saved_h, slabs, group)
        slab_fwd_stage(plan, params, slabs, 2, x, out, h0, saved_h, saved_h1, saved_p, ws, stream)
        ctx.plan, ctx.slabs, ctx.act, ctx.skip, ctx.group = plan, slabs, act, skip, group
        ctx.reduce_grads = reduce_grads
        ctx.save_for_backward(x, w_down, b_down, w_up, b_up, scalar, full)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x, w_down, b_down, w_up, b_up, scalar, full = ctx.saved_tensors
        plan, slabs, act, group = ctx.plan, ctx.slabs, ctx.act, ctx.group
        g = grad_out.contiguous()
        _require_hip_f32(g, "grad_output")
        n_loc, d = x.shape
        r = w_down.shape[0]
        fp = full.shape[2]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        saved_h, saved_p = full[1], full[2]
        saved_h1 = full[3] if act == 2 else None
        ex = torch.empty(2, slabs.padded, fp, dtype=torch.float32, device=dev)
        dp, dh1 = ex[0], ex[1]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        flat, (gwd, gbd, gwu, gbu, gs) = _flat_param_grads(w_down, b_down, w_up, b_up, scalar)
        lo, hi = slabs.bounds()
        ws = torch.empty(max(int(lib.gconv_adapter_slab_bwd_workspace_bytes(plan.ref(), lo, hi, d, r)), 16),
                         dtype=torch.uint8, device=dev)
        params = _lib.AdapterParams(w_down.data_ptr(), _ptr(b_down), w_up.data_ptr(), _ptr(b_up),
                                    _ptr(scalar), d, r, act, int(ctx.skip))
        stream = _stream(dev.index)
        args = (x, g, saved_h, saved_h1, saved_p, dp, dh1, gx, gwd, gbd, gwu, gbu, gs, ws, stream)
        slab_bwd_stage(plan, params, slabs, 0, *args)
        exchange_rows(dp, slabs, group)
        slab_bwd_stage(plan, params, slabs, 1, *args)
        exchange_rows(dh1, slabs, group)
        slab_bwd_stage(plan, params, slabs, 2, *args)
        if ctx.reduce_grads and slabs.world > 1:   # every rank's rows contribute to every gradient
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return gx, gwd, gbd, gwu, gbu, gs, None, None, None, None, None, None


class RowSlabGConvAdapter(GConvAdapter):
    """``GConvAdapter`` for one big graph partitioned by rows over the ranks of a process group.

    Same constructor, parameters and ``state_dict`` as ``GConvAdapter`` (the reference's
    module surface, ``src/finetune/gconv_adapter.py:23-108``); call ``partition(num_nodes)``
    once, then ``forward(x_rows, edge_index)`` with THIS rank's rows of x and the FULL edge
    list. Returns this rank's rows of the output.

    Parameter gradients: every rank's rows contribute to every trainable parameter (the
    adapter's, a norm layer's affine, the head), so after ``backward()`` each ``p.grad`` holds
    this rank's CONTRIBUTION and the caller sums them over the ranks once per step, all
    parameters in one collective (``dist.FlatGradBucket.all_reduce_grads()`` — the same single
    exchange the graph-sharded molecule path uses). ``reduce_param_grads=True`` instead sums the
    five adapter gradients inside backward (one small all-reduce per adapter; stand-alone use).
    ``layer_norm`` is row-local and exact; ``batch_norm`` would use per-rank statistics (the
    reference's transductive inputs are [1, N, d], where it is skipped: ``:98-100``)."""

    #: a copy (deepcopy / pickle / torch.save) keeps its partition only if it is default-group based:
    #: a process-group handle cannot travel, so the group is dropped and the copy re-partitions
    _TRANSIENT = GConvAdapter._TRANSIENT + ("_group",)

    def __getstate__(self):
        state = super().__getstate__()
        if self.__dict__.get("_group") is not None:   # partition of a sub-group: call partition() again
            state.pop("_slabs", None)
            state.pop("_reduce_grads", None)
        return state

    def partition(self, num_nodes: int, group=None, world: Optional[int] = None,
                  rank: Optional[int] = None, reduce_param_grads: bool = False) -> RowSlabs:
        if world is None:
            world = dist.get_world_size(group) if _active(group) else 1
        if rank is None:
            rank = dist.get_rank(group) if _active(group) else 0
        self.__dict__["_slabs"] = RowSlabs(num_nodes, world, rank)
        self.__dict__["_group"] = group
        self.__dict__["_reduce_grads"] = bool(reduce_param_grads)
        return self.__dict__["_slabs"]

    def _forward_2d(self, x2d: torch.Tensor, edge_index: torch.Tensor, fuse_scalar: bool) -> torch.Tensor:
        slabs = self.__dict__.get("_slabs")
        if slabs is None:
            raise RuntimeError("RowSlabGConvAdapter: call partition(num_nodes) before forward")
        plan = get_plan(edge_index, slabs.num_nodes, self.normalize)
        return RowSlabAdapterFunction.apply(
            x2d, self.conv_down.lin.weight, self.conv_down.bias, self.conv_up.lin.weight,
            self.conv_up.bias, self.scalar if fuse_scalar else None, plan, slabs,
            _lib.ACT_CODES[self.non_linearity], bool(self.skip_connection), self.__dict__.get("_group"),
            self.__dict__.get("_reduce_grads", False))
