"""BatchNorm1d over node features on the device (``csrc/norm.hip``).

``BatchNorm1d`` is an ``nn.BatchNorm1d`` — same constructor, parameters, buffers and ``state_dict``
keys, so the reference's ``batch_norm_layers`` (``/root/reference/src/models/inductive/gnn.py:47-48,
108-110``) and a pretrained checkpoint load unchanged — whose forward / backward on ``[N, d]`` fp32
HIP tensors run the three-launch kernels of ``gconv_batch_norm_*`` instead of torch's channels-last
BatchNorm kernels (72 + 10 us forward, 88 + 15 us backward on a 512-molecule batch — 28 % of the
whole fine-tune step). ``fused_relu=True`` folds in the ReLU that follows the norm in every layer but
the last (``gnn.py:127-128``): one pass less each way.

Sharded batches (one process per GPU, a shard of the batch's graphs each): ``process_group`` makes
the layer normalise with the statistics of the WHOLE batch, as the single-process reference does
(``gnn.py:56,109-110``; the adapter's own tail ``gconv_adapter.py:98-100``) — one all-gather of
``[2 d + 4]`` floats per rank forward, one all-reduce of ``[2 d]`` floats backward, both between
launches of the same kernels (``gconv_batch_norm_{local_stats,fwd_sync,bwd_local,bwd_dx}_f32``).
``sync_batch_norm(model, group)`` switches every ``BatchNorm1d`` of a model over.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, _torch_ext
from .functional import _ptr, _require_hip_f32, _stream


class BatchNormFunction(torch.autograd.Function):
    """y = [relu](gamma * (x - mean_batch) * rstd_batch + beta); running statistics updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum: float, eps: float, relu: bool):
        lib = _lib.load()
        _require_hip_f32(x, "x")
        if x.dim() != 2:
            raise RuntimeError(f"BatchNormFunction takes [N, d] node features, got {tuple(x.shape)}")
        x = x.contiguous()
        n, d = x.shape
        if n < 2:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        y = torch.empty_like(x)
        stats = torch.empty(2, d, dtype=torch.float32, device=dev)
        nbytes = int(lib.gconv_batch_norm_workspace_bytes(n, d))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        st = lib.gconv_batch_norm_fwd_f32(x.data_ptr(), _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                          _ptr(running_var), float(momentum), float(eps), int(relu),
                                          y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), n, d,
                                          ws.data_ptr(), ws.numel(), _stream(dev.index))
        if st != 0:
            _lib.check(st, "batch_norm_fwd")
        ctx.relu = relu
        ctx.save_for_backward(x, y if relu else None, stats, gamma)
        ctx.has_beta = beta is not None
        return y   # (the running statistics are buffers: updated in place, outside autograd)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, y, stats, gamma = ctx.saved_tensors
        g = g.contiguous()
        n, d = x.shape
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ggam = torch.empty(d, dtype=torch.float32, device=dev) if gamma is not None else None
        gbet = torch.empty(d, dtype=torch.float32, device=dev) if ctx.has_beta else None
        nbytes = int(lib.gconv_batch_norm_workspace_bytes(n, d))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        st = lib.gconv_batch_norm_bwd_f32(g.data_ptr(), x.data_ptr(), _ptr(y), stats[0].data_ptr(),
                                          stats[1].data_ptr(), _ptr(gamma), _ptr(gx), _ptr(ggam), _ptr(gbet),
                                          n, d, ws.data_ptr(), ws.numel(), _stream(dev.index))
        if st != 0:
            _lib.check(st, "batch_norm_bwd")
        return gx, ggam, gbet, None, None, None, None, None


def _all_gather_flat(local: torch.Tensor, world: int, group) -> torch.Tensor:
    """``[world * len(local)]``: every rank's vector in rank order. One RCCL call on the nccl backend;
    gloo (tests, rehearsals on one GPU) stages through the host."""
    if dist.get_backend(group) == "nccl":
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    host = local.cpu()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host, group=group)
    return torch.cat(parts).to(local.device)


def _all_reduce_sum(t: torch.Tensor, group) -> torch.Tensor:
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t
    host = t.cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    return host.to(t.device)


class SyncBatchNormFunction(torch.autograd.Function):
    """``BatchNormFunction`` over a batch whose rows are sharded across the ranks of ``group``: mean /
    variance / running statistics of ALL rows, merged in rank order (the same bits on every rank);
    d x with the all-reduced column sums. d gamma / d beta are this rank's CONTRIBUTIONS — summed by
    the step's gradient all-reduce like every other parameter gradient (``dist.FlatGradBucket``)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum: float, eps: float, relu: bool, group):
        lib = _lib.load()
        _require_hip_f32(x, "x")
        if x.dim() != 2:
            raise RuntimeError(f"SyncBatchNormFunction takes [N, d] node features, got {tuple(x.shape)}")
        x = x.contiguous()
        n, d = x.shape
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        world = dist.get_world_size(group)
        words = int(lib.gconv_batch_norm_sync_words(d))
        local = torch.empty(words, dtype=torch.float32, device=dev)
        ws = torch.empty(max(int(lib.gconv_batch_norm_workspace_bytes(n, d)), 16), dtype=torch.uint8, device=dev)
        stream = _stream(dev.index)
        _lib.check(lib.gconv_batch_norm_local_stats_f32(_ptr(x) if n else None, n, d, local.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), stream), "batch_norm_local_stats")
        gathered = _all_gather_flat(local, world, group)
        y = torch.empty_like(x)
        stats = torch.empty(2, d, dtype=torch.float32, device=dev)
        total = torch.empty(4, dtype=torch.int32, device=dev)
        _lib.check(lib.gconv_batch_norm_fwd_sync_f32(_ptr(x) if n else None, _ptr(gamma), _ptr(beta),
                                                     _ptr(running_mean), _ptr(running_var), float(momentum),
                                                     float(eps), int(relu), _ptr(y) if n else None,
                                                     stats[0].data_ptr(), stats[1].data_ptr(), total.data_ptr(),
                                                     n, d, gathered.data_ptr(), world, stream), "batch_norm_fwd_sync")
        ctx.relu, ctx.group, ctx.has_beta = relu, group, beta is not None
        ctx.save_for_backward(x, y if relu else None, stats, gamma, total)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, y, stats, gamma, total = ctx.saved_tensors
        g = g.contiguous()
        n, d = x.shape
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = _stream(dev.index)
        sums = torch.empty(2 * d, dtype=torch.float32, device=dev)
        ws = torch.empty(max(int(lib.gconv_batch_norm_workspace_bytes(n, d)), 16), dtype=torch.uint8, device=dev)
        _lib.check(lib.gconv_batch_norm_bwd_local_f32(_ptr(g) if n else None, _ptr(x) if n else None,
                                                      _ptr(y) if n else None, stats[0].data_ptr(), stats[1].data_ptr(),
                                                      sums.data_ptr(), n, d, ws.data_ptr(), ws.numel(), stream),
                   "batch_norm_bwd_local")
        gx = None
        if ctx.needs_input_grad[0]:
            sums_all = _all_reduce_sum(sums.clone(), ctx.group)
            gx = torch.empty_like(x)
            _lib.check(lib.gconv_batch_norm_bwd_dx_f32(_ptr(g) if n else None, _ptr(x) if n else None,
                                                       _ptr(y) if n else None, stats[0].data_ptr(),
                                                       stats[1].data_ptr(), _ptr(gamma), sums_all.data_ptr(),
                                                       total.data_ptr(), _ptr(gx) if n else None, n, d, stream),
                       "batch_norm_bwd_dx")
        ggam = sums[:d] if gamma is not None else None
        gbet = sums[d:] if ctx.has_beta else None
        return gx, ggam, gbet, None, None, None, None, None, None


def sync_batch_norm(module: nn.Module, process_group=None) -> nn.Module:
    """Every ``gconv_adapter_amd.BatchNorm1d`` under ``module`` (the backbone's ``batch_norm_layers``,
    the adapters' ``normalization='batch_norm'`` tails) normalises with the statistics of the batch of
    ALL ranks of ``process_group`` (default: the world group) from now on. Returns ``module``.
    The counterpart of ``torch.nn.SyncBatchNorm.convert_sync_batchnorm`` that keeps the HIP kernels."""
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("sync_batch_norm needs an initialised process group")
    group = process_group if process_group is not None else dist.group.WORLD
    for m in module.modules():
        if isinstance(m, BatchNorm1d):
            m.process_group = group
    return module


def batch_norm_eval(x, running_mean, running_var, gamma, beta, eps: float, relu: bool) -> torch.Tensor:
    """Evaluation-mode forward (running statistics; autograd through the torch formula when needed)."""
    if torch.is_grad_enabled() and (x.requires_grad or (gamma is not None and gamma.requires_grad)):
        y = torch.nn.functional.batch_norm(x, running_mean, running_var, gamma, beta, False, 0.0, eps)
        return torch.relu(y) if relu else y
    lib = _lib.load()
    _require_hip_f32(x, "x")
    x = x.contiguous()
    n, d = x.shape
    rstd = torch.rsqrt(running_var + eps)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = lib.gconv_batch_norm_apply_f32(x.data_ptr(), running_mean.data_ptr(), rstd.data_ptr(), _ptr(gamma),
                                            _ptr(beta), int(relu), y.data_ptr(), n, d, _stream())
    _lib.check(st, "batch_norm_apply")
    return y


class BatchNorm1d(nn.BatchNorm1d):
    """``nn.BatchNorm1d`` with the HIP kernels for ``[N, d]`` fp32 inputs on a HIP device (module
    docstring). Anything else (3-D input, other dtypes, ``d`` not a multiple of 4 or > 1024) takes
    ``nn.BatchNorm1d``'s own path, so the class can replace it everywhere."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: Optional[float] = 0.1,
                 affine: bool = True, track_running_stats: bool = True, device=None, dtype=None,
                 fused_relu: bool = False):
        super().__init__(num_features, eps, momentum, affine, track_running_stats, device=device, dtype=dtype)
        self.fused_relu = fused_relu
        #: a torch.distributed group: training-mode statistics over the rows of all its ranks
        #: (``sync_batch_norm``); None = this process's rows, as ``nn.BatchNorm1d``
        self.process_group = None

    def __getstate__(self):   # (a process group handle does not travel with a copy / a checkpoint)
        state = self.__dict__.copy()
        state["process_group"] = None
        return state

    def _native(self, x: torch.Tensor, allow_empty: bool = False) -> bool:
        return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 == 0
                and x.shape[1] <= 1024 and (x.shape[0] >= 1 or allow_empty))

    def forward(self, x: torch.Tensor, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``scale`` (a one-element tensor, the adapter's learnable scalar): returns ``scale * BN(x)`` in the
        same three launches — ``s (gamma xhat + beta) = (s gamma) xhat + (s beta)``, so the kernels run
        with the scaled affine pair and autograd's chain rule through the two ``[d]``-sized products
        yields d gamma, d beta and d scale; no extra pass over ``[N, d]`` in either direction (the
        reference multiplies the whole output, ``gconv_adapter.py:105-106``)."""
        gamma, beta = self.weight, self.bias
        if scale is not None:
            if self.fused_relu or gamma is None or beta is None or not self._native(x):
                return self.forward(x) * scale
            gamma, beta = gamma * scale, beta * scale
        group = self.__dict__.get("process_group")
        synced = group is not None and self.training and dist.get_world_size(group) > 1
        if not self._native(x) and not (synced and self._native(x, allow_empty=True)):
            if synced:
                raise RuntimeError("gconv_adapter_amd.BatchNorm1d with a process group takes [N, d] fp32 HIP "
                                   f"tensors (d a multiple of 4, <= 1024), got {x.dtype} {tuple(x.shape)} on {x.device}")
            y = super().forward(x)
            return torch.relu(y) if self.fused_relu else y
        use_batch_stats = self.training or self.running_mean is None
        if not use_batch_stats:
            return batch_norm_eval(x, self.running_mean, self.running_var, gamma, beta, self.eps,
                                   self.fused_relu)
        ext = None if synced else _torch_ext.get()
        counts = self.training and self.track_running_stats and self.num_batches_tracked is not None
        # num_batches_tracked += 1: by the C++ node's reduction launch when it can be (a launch of torch's per
        # layer otherwise); momentum=None (cumulative average) needs the count on the host first
        in_kernel = (counts and ext is not None and self.momentum is not None and x.shape[0] >= 2
                     and self.num_batches_tracked.is_cuda)
        factor = 0.0
        if counts:
            if not in_kernel:
                self.num_batches_tracked.add_(1)
            factor = (1.0 / float(self.num_batches_tracked)) if self.momentum is None else self.momentum
        rm = self.running_mean if self.training and self.track_running_stats else None
        rv = self.running_var if self.training and self.track_running_stats else None
        if synced:
            return SyncBatchNormFunction.apply(x, gamma, beta, rm, rv, factor, self.eps, self.fused_relu, group)
        if ext is None:
            return BatchNormFunction.apply(x, gamma, beta, rm, rv, factor, self.eps, self.fused_relu)
        # the same node in C++ (csrc/torch_binding.cpp BatchNormFn): one host call forward, none from Python backward
        if x.shape[0] < 2:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
        if x.device.index != torch.cuda.current_device():
            torch.cuda.set_device(x.device)
        return ext.batch_norm(x, gamma, beta, rm, rv, self.num_batches_tracked if in_kernel else None, factor,
                              self.eps, self.fused_relu, _stream(x.device.index))
