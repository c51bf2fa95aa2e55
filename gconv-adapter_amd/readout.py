"""The end of the fine-tune step's forward pass on the device (``csrc/readout.hip``).

``global_mean_pool`` / ``global_add_pool`` take PyG's arguments (``x, batch, size=None``) — what the reference's
graph-level model calls as ``self.pool(h, batch)`` (``/root/reference/src/models/inductive/gnn.py:187-195``) —
and ``masked_loss`` is the training loop's NaN-masked fp64 loss
(``/root/reference/scripts/finetune_inductive_learning.py:50-63``) as ONE launch forward and a scaling backward.
On a 512-molecule shard torch spends ~25 small launches (4 - 8 us each, on the device and on the host) on
the two.

``batch`` must be sorted ascending, as PyG's collation emits it. The pooling kernel checks that on its way: graphs
whose rows are out of order come out as NaN rows (visible in the loss without a host synchronisation), and
``validate=True`` raises instead (one synchronisation); ``size=None`` reads ``batch.max()`` back, as PyG does —
pass the number of graphs.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .functional import _require_hip_f32, _stream


def _on_device(t: torch.Tensor) -> None:
    if t.device.index != torch.cuda.current_device():
        torch.cuda.set_device(t.device)


class SegmentPoolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, batch, size: int, mean: bool):
        lib = _lib.load()
        _require_hip_f32(x, "x")
        if x.dim() != 2 or batch.dim() != 1 or batch.shape[0] != x.shape[0]:
            raise RuntimeError(f"global pool takes x [N, d] and batch [N], got {tuple(x.shape)} and {tuple(batch.shape)}")
        if batch.dtype != torch.int64 or batch.device != x.device:
            raise RuntimeError("batch must be an int64 tensor on x's device (PyG convention)")
        x, batch = x.contiguous(), batch.contiguous()
        _on_device(x)
        n, d = x.shape
        out = torch.empty(size, d, dtype=torch.float32, device=x.device)
        counts = torch.empty(size + 1, dtype=torch.int32, device=x.device)   # (+ the "batch not sorted" flag word)
        st = lib.gconv_segment_pool_f32(x.data_ptr(), batch.data_ptr(), n, d, size, int(mean), out.data_ptr(),
                                        counts.data_ptr(), _stream(x.device.index))
        if st != 0:
            _lib.check(st, "segment_pool")
        ctx.mean, ctx.size = mean, size
        ctx.save_for_backward(batch, counts)
        ctx.mark_non_differentiable(counts)
        return out, counts

    @staticmethod
    def backward(ctx, go, _gcounts=None):
        lib = _lib.load()
        batch, counts = ctx.saved_tensors
        go = go.contiguous()
        _on_device(go)
        n, d = batch.shape[0], go.shape[1]
        gx = torch.empty(n, d, dtype=torch.float32, device=go.device)
        st = lib.gconv_segment_pool_bwd_f32(go.data_ptr(), batch.data_ptr(), counts.data_ptr(), n, d, ctx.size,
                                            int(ctx.mean), gx.data_ptr(), _stream(go.device.index))
        if st != 0:
            _lib.check(st, "segment_pool_bwd")
        return gx, None, None, None


def _pool(x: torch.Tensor, batch: Optional[torch.Tensor], size: Optional[int], mean: bool, validate: bool):
    if batch is None:   # PyG: one graph
        return x.mean(0, keepdim=True) if mean else x.sum(0, keepdim=True)
    if not x.is_cuda:
        raise RuntimeError("gconv_adapter_amd has no CPU path (x is on %s)" % x.device)
    if size is None:
        size = int(batch.max()) + 1 if batch.numel() else 0
    native = x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and x.shape[1] <= 1024
    if not native:      # other widths / dtypes: torch's scatter, same semantics
        out = torch.zeros(size, *x.shape[1:], dtype=x.dtype, device=x.device).index_add_(0, batch, x)
        if mean:
            cnt = torch.zeros(size, dtype=x.dtype, device=x.device).index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
            out = out / cnt.clamp(min=1).view(-1, *([1] * (x.dim() - 1)))
        return out
    out, counts = SegmentPoolFunction.apply(x, batch, int(size), mean)
    # The kernel checks the order on its way (csrc/readout.hip): a batch vector that is not sorted, or holds ids
    # outside [0, size), gives NaN rows — the loss shows it — and raises counts[size]; `validate=True` reads the flag
    # back (one host synchronisation) and raises here instead.
    if validate and int(size) > 0 and int(counts[int(size)]) != 0:
        raise RuntimeError("global pool: `batch` must be sorted ascending with ids in [0, size) (PyG's collation "
                           "order); the affected graphs' rows were set to NaN")
    return out


def global_mean_pool(x: torch.Tensor, batch: Optional[torch.Tensor], size: Optional[int] = None,
                     validate: bool = False) -> torch.Tensor:
    """PyG's ``global_mean_pool(x, batch, size=None)``: row ``g`` = mean of the rows of graph ``g``."""
    return _pool(x, batch, size, True, validate)


def global_add_pool(x: torch.Tensor, batch: Optional[torch.Tensor], size: Optional[int] = None,
                    validate: bool = False) -> torch.Tensor:
    """PyG's ``global_add_pool(x, batch, size=None)``: row ``g`` = sum of the rows of graph ``g``."""
    return _pool(x, batch, size, False, validate)


#: up to this many logits ONE workgroup computes the loss (HIV: 4096 x 1); beyond it (ToxCast: 617 tasks x 4096)
#: gconv_masked_loss_ws_f64 spreads the fp64 exp / log1p over 256 workgroups
_LOSS_ONE_WORKGROUP = 16384
_LOSS_WS = {}


def _loss_workspace(lib, device) -> torch.Tensor:
    """One zero-initialised workspace per device (the kernels leave its ticket word at zero; launches on one device
    are ordered by the stream the training loop runs on)."""
    ws = _LOSS_WS.get(device.index)
    if ws is None:
        ws = torch.zeros(int(lib.gconv_masked_loss_workspace_bytes(0)), dtype=torch.uint8, device=device)
        _LOSS_WS[device.index] = ws
    return ws


class MaskedLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, kind: int):
        lib = _lib.load()
        _require_hip_f32(output, "output")
        if target.dtype != torch.float64 or target.device != output.device or target.numel() != output.numel():
            raise RuntimeError("masked_loss: target must be an fp64 tensor of output's size on its device "
                               "(NaN = no label)")
        o, t = output.contiguous(), target.contiguous()
        _on_device(o)
        res = torch.empty(2, dtype=torch.float64, device=o.device)
        grad = torch.empty_like(o)
        if o.numel() > _LOSS_ONE_WORKGROUP:   # a multi-task head: block partials in a fixed order, two launches
            ws = _loss_workspace(lib, o.device)
            st = lib.gconv_masked_loss_ws_f64(o.data_ptr(), t.data_ptr(), o.numel(), kind, res.data_ptr(),
                                              grad.data_ptr(), ws.data_ptr(), ws.numel(), _stream(o.device.index))
        else:
            st = lib.gconv_masked_loss_f64(o.data_ptr(), t.data_ptr(), o.numel(), kind, res.data_ptr(), grad.data_ptr(),
                                           _stream(o.device.index))
        if st != 0:
            _lib.check(st, "masked_loss")
        ctx.save_for_backward(grad)
        ctx.shape = output.shape
        return res[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g.to(torch.float32)).view(ctx.shape), None, None


def masked_loss(output: torch.Tensor, target: torch.Tensor, kind: str = "bce") -> torch.Tensor:
    """``sum(criterion(output[valid].double(), target[valid])) / valid.sum()`` with ``valid = ~isnan(target)``
    (``finetune_inductive_learning.py:56-63``); ``kind``: ``'bce'`` (``BCEWithLogitsLoss``) or ``'mse'``
    (``MSELoss``). An fp64 scalar on the device; no valid target gives 0 (the reference skips such a batch,
    which costs it a host synchronisation per step)."""
    if kind not in ("bce", "mse"):
        raise ValueError(f"masked_loss: kind must be 'bce' or 'mse', got {kind!r}")
    if not output.is_cuda:
        raise RuntimeError("gconv_adapter_amd has no CPU path (output is on %s)" % output.device)
    return MaskedLossFunction.apply(output, target.view(output.shape), 0 if kind == "bce" else 1)
