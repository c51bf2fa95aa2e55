"""Fused Adam over the trainable set.

Drop-in for the ``torch.optim.Adam(param_groups, lr=..., weight_decay=...)`` of the reference's
fine-tuning loops (``/root/reference/scripts/finetune_inductive_learning.py:375-379``,
``scripts/finetune_transductive_learning.py:205``): same constructor arguments (per-group
``lr`` / ``weight_decay`` / ``betas`` / ``eps``), same update rule (the step count of the bias
correction is kept PER PARAMETER, as torch does: a parameter without a gradient is skipped and its
count does not advance), ``state_dict()`` in the same per-parameter format — but the trainable parameters of a group live in ONE flat buffer
(``p.data`` of every parameter becomes a view of it; ``Parameter`` objects, names and
``state_dict`` keys of the model are untouched) and a step is ONE multi-tensor copy of the
gradients plus ONE ``gconv_adam_step_f32`` launch per group instead of ~10 multi-tensor
launches. Frozen parameters (``requires_grad=False``) are ignored, as torch's Adam ignores
parameters without a gradient. Every parameter starts on a 16-byte boundary of the flat buffer
(the adapter kernels read weights and biases with 16-byte loads and reject misaligned ones —
a ``scalar`` of one float would otherwise shift everything behind it); the padding floats are
zero, receive zero gradients and stay zero. There is no CPU path.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import _lib
from .functional import _stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._lib = _lib.load()
        self._flat: List[dict] = []
        for group in self.param_groups:
            self._flat.append(self._flatten(group))

    # ------------------------------------------------------------------ construction
    def _flatten(self, group: dict) -> dict:
        ps = [p for p in group["params"] if p.requires_grad]
        if not ps:
            return dict(params=[], total=0)
        dev, dt = ps[0].device, ps[0].dtype
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("FusedAdam needs fp32 parameters on one HIP device (no CPU path)")
        total = sum(-(-p.numel() // 4) * 4 for p in ps)
        flat = torch.zeros(total, dtype=dt, device=dev)
        m = torch.zeros_like(flat)
        v = torch.zeros_like(flat)
        off = 0
        offsets = []
        with torch.no_grad():
            for p in ps:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view_as(p)          # same Parameter, storage moved
                self.state[p] = {"step": torch.zeros((), dtype=torch.float32),
                                 "exp_avg": m[off:off + n].view_as(p),
                                 "exp_avg_sq": v[off:off + n].view_as(p)}
                offsets.append(off)
                off += -(-n // 4) * 4
        grad = torch.zeros_like(flat)
        grad_views = [grad[o:o + p.numel()].view_as(p) for p, o in zip(ps, offsets)]
        # the gradient views double as the parameters' gradient SINKS: the adapter's backward node
        # writes dW / db / ds straight into `grad` (csrc/torch_binding.cpp, set_grad_sink) and step()
        # finds them in place — no copy at all for the adapters, one multi-tensor copy for the rest
        from . import _torch_ext
        ext = _torch_ext.get()
        if ext is not None and hasattr(ext, "set_grad_sink"):
            for p, gv in zip(ps, grad_views):
                ext.set_grad_sink(p, gv)
        return dict(params=ps, offsets=offsets, total=total, flat=flat, grad=grad, grad_views=grad_views,
                    exp_avg=m, exp_avg_sq=v, steps=[0] * len(ps))

    def flat_grads(self) -> List[torch.Tensor]:
        """The flat gradient buffer of every parameter group (what a data-parallel loop all-reduces,
        after ``gather_grads()``): padding floats are zero."""
        return [fl["grad"] for fl in self._flat if fl.get("params")]

    @torch.no_grad()
    def gather_grads(self) -> None:
        """Brings every gradient that is not in its group's flat buffer yet into it (one multi-tensor
        copy per group) and re-binds ``p.grad`` to the buffer's view."""
        for fl in self._flat:
            src, dst = [], []
            for p, gv in zip(fl.get("params", []), fl.get("grad_views", [])):
                g = p.grad
                if g is not None and g.data_ptr() != gv.data_ptr():
                    src.append(g)
                    dst.append(gv)
                    p.grad = gv
            if src:
                torch._foreach_copy_(dst, src)

    # ------------------------------------------------------------------ data parallelism
    def all_reduce_grads(self, average: bool = False, async_op: bool = False, group=None):
        """The step's gradient exchange when the parameters live in this optimizer (the flat
        buffer Adam reads IS the all-reduce bucket): gradients that are not in the flat buffers yet are
        packed (``gather_grads``: one multi-tensor copy per group; the adapters' arrive in place through the
        gradient sinks), then ONE collective per parameter group on its flat gradient buffer. A parameter
        WITHOUT a gradient on this rank (an unused branch) contributes zeros and gets the other ranks' sum
        — every rank must make the same call. ``async_op``: returns at once; ``wait()`` (``step()`` calls
        it) orders the current stream behind the collectives and applies ``average``. No-op without an
        initialised process group or with one rank."""
        import torch.distributed as dist
        self.wait()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        self.gather_grads()
        works = []
        with torch.no_grad():
            for fl in self._flat:
                if not fl.get("params"):
                    continue
                for p, gv in zip(fl["params"], fl["grad_views"]):
                    if p.grad is None:            # (the flat buffer may still hold an older step's values there)
                        gv.zero_()
                        p.grad = gv
                works.append(dist.all_reduce(fl["grad"], op=dist.ReduceOp.SUM, group=group, async_op=async_op))
        world = dist.get_world_size(group)
        if async_op:
            self._pending = (works, world if average else 0)
            return works
        if average:
            for fl in self._flat:
                if fl.get("params"):
                    fl["grad"].div_(world)
        return works

    def wait(self) -> None:
        """Orders the current stream behind an ``all_reduce_grads(async_op=True)`` (and divides for
        ``average=True``)."""
        pending = self.__dict__.pop("_pending", None)
        if pending is None:
            return
        works, div = pending
        for w in works:
            w.wait()
        if div:
            for fl in self._flat:
                if fl.get("params"):
                    fl["grad"].div_(div)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        self.wait()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, fl in zip(self.param_groups, self._flat):
            ps = fl["params"]
            if not ps:
                continue
            grads = [p.grad for p in ps]
            has = [g is not None for g in grads]
            if not any(has):
                continue
            # torch.optim.Adam counts the step PER PARAMETER (a parameter without a gradient is
            # skipped and its bias correction does not advance): keep one counter each
            steps = fl["steps"]
            for i, h in enumerate(has):
                if h:
                    steps[i] += 1
            beta1, beta2 = group["betas"]
            dev = fl["flat"].device
            if all(has) and min(steps) == max(steps):   # the normal case: (one multi-tensor copy,) one launch
                pend = [(gv, g) for gv, g in zip(fl["grad_views"], grads) if g.data_ptr() != gv.data_ptr()]
                if pend:   # gradients the backward pass has not already deposited in the flat buffer
                    torch._foreach_copy_([a for a, _ in pend], [b for _, b in pend])
                st = self._lib.gconv_adam_step_f32(
                    fl["flat"].data_ptr(), fl["grad"].data_ptr(), fl["exp_avg"].data_ptr(),
                    fl["exp_avg_sq"].data_ptr(), fl["total"], float(group["lr"]), float(beta1), float(beta2),
                    float(group["eps"]), float(group["weight_decay"]), steps[0], _stream(dev.index))
                _lib.check(st, "adam_step")
                continue
            # contiguous runs of parameters that have a gradient AND the same step count: one
            # launch per run, each with its own bias correction
            i = 0
            while i < len(ps):
                if not has[i]:
                    i += 1
                    continue
                j = i
                while j < len(ps) and has[j] and steps[j] == steps[i]:
                    j += 1
                lo = fl["offsets"][i]
                hi = fl["offsets"][j - 1] + ps[j - 1].numel()
                pend = [(gv, g) for gv, g in zip(fl["grad_views"][i:j], grads[i:j]) if g.data_ptr() != gv.data_ptr()]
                if pend:
                    torch._foreach_copy_([a for a, _ in pend], [b for _, b in pend])
                st = self._lib.gconv_adam_step_f32(
                    fl["flat"].data_ptr() + 4 * lo, fl["grad"].data_ptr() + 4 * lo,
                    fl["exp_avg"].data_ptr() + 4 * lo, fl["exp_avg_sq"].data_ptr() + 4 * lo, hi - lo,
                    float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                    float(group["weight_decay"]), steps[i], _stream(dev.index))
                _lib.check(st, "adam_step")
                i = j
        return loss

    # ------------------------------------------------------------------ (de)serialisation
    def state_dict(self):
        """``torch.optim.Adam``-compatible: per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``
        (the moments are views of the flat buffers: deep-copy before loading it elsewhere, as with
        any optimizer state dict that is not going through ``torch.save``)."""
        for fl in self._flat:
            for p, k in zip(fl.get("params", []), fl.get("steps", [])):
                self.state[p]["step"] = torch.tensor(float(k))
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        """Accepts a ``torch.optim.Adam`` / ``FusedAdam`` state dict: moments are copied INTO the
        flat buffers (the per-parameter state tensors stay views of them)."""
        views = {id(p): dict(self.state[p]) for fl in self._flat for p in fl.get("params", [])}
        super().load_state_dict(state_dict)
        for group, fl in zip(self.param_groups, self._flat):
            for p in fl.get("params", []):
                loaded, mine = self.state.get(p, {}), views[id(p)]
                if loaded and loaded.get("exp_avg") is not mine["exp_avg"]:
                    mine["exp_avg"].copy_(loaded["exp_avg"])
                    mine["exp_avg_sq"].copy_(loaded["exp_avg_sq"])
                    mine["step"] = torch.as_tensor(float(loaded["step"]))
                    self.state[p] = mine
                elif not loaded:   # torch's Adam keeps no state for a parameter that never had a gradient:
                    mine["exp_avg"].zero_()          # it starts from zero moments at step 0
                    mine["exp_avg_sq"].zero_()
                    mine["step"] = torch.zeros((), dtype=torch.float32)
                    self.state[p] = mine
            if fl.get("params"):
                fl["steps"] = [int(self.state[p]["step"]) for p in fl["params"]]


def zero_grad(params, set_to_none: bool = True) -> None:
    """``optimizer.zero_grad(set_to_none=True)`` of the reference's loop (``finetune_inductive_learning.py:49``) for a
    parameter list as ONE call into the C++ binding (0.3 us per parameter in Python: 7 us of a 155 us ESOL step);
    falls back to the Python loop without the binding or with ``set_to_none=False``."""
    from . import _torch_ext
    ext = _torch_ext.get()
    if set_to_none and ext is not None and hasattr(ext, "clear_grads"):
        ext.clear_grads(params if isinstance(params, list) else list(params))
        return
    for p in params:
        if set_to_none:
            p.grad = None
        elif p.grad is not None:
            p.grad.zero_()
