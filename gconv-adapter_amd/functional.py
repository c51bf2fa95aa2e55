"""Autograd bridge: ``torch.autograd.Function`` over the C ABI, plus thin wrappers
of the stand-alone kernels (SpMM, projections) for tests and micro-benchmarks.

PyTorch is plumbing here (device memory, the current HIP stream, autograd
book-keeping); every FLOP of the adapter path runs in ``libgconv_adapter.so``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .graph import GraphPlan


def _require_hip_f32(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"gconv_adapter_amd has no CPU path: `{name}` must live on a HIP device")
    if t.dtype != torch.float32:
        raise RuntimeError(f"gconv_adapter_amd computes in fp32: `{name}` has dtype {t.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device_index: Optional[int] = None) -> int:
    """Raw hipStream_t of torch's current stream (the C getter is ~20x cheaper than
    ``torch.cuda.current_stream().cuda_stream``, which matters on launch-bound batches)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)
    return torch.cuda.current_stream().cuda_stream


def _ptr(t) -> Optional[int]:
    return None if t is None else t.data_ptr()


def adapter_infer(x, w_down, b_down, w_up, b_up, scalar, plan: GraphPlan, act: int, skip: bool,
                  walk: int = 0) -> torch.Tensor:
    """Forward only (``gconv_adapter_infer_f32``): what ``GConvAdapter.forward`` runs under ``torch.no_grad()``
    or when nothing that reaches it requires a gradient — the reference's evaluation passes
    (``/root/reference/scripts/finetune_inductive_learning.py:95-104``). Same bits in ``out`` as the training
    forward; no saved tensors are allocated or written."""
    lib = _lib.load()
    if not x.is_cuda or x.dtype != torch.float32:
        _require_hip_f32(x, "x")
    n, d = x.shape
    r = w_down.shape[0]
    if n != plan.num_nodes:
        raise RuntimeError(f"x has {n} nodes but the graph plan was built for {plan.num_nodes}")
    plan.sizes(lib, d, r)   # (validates the rank)
    ws_bytes = plan.infer_bytes(lib, d, r)
    x, w_down, w_up = x.contiguous(), w_down.contiguous(), w_up.contiguous()
    dev = x.device
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    out = torch.empty_like(x)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    params = _lib.AdapterParams(w_down.data_ptr(), _ptr(b_down), w_up.data_ptr(), _ptr(b_up),
                                _ptr(scalar), d, r, act, int(skip), int(walk))
    st = lib.gconv_adapter_infer_f32(plan.ref(), C.byref(params), x.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                     ws_bytes, _stream(dev.index))
    if st != 0:
        _lib.check(st, "adapter_infer")
    return out


class GConvAdapterFunction(torch.autograd.Function):
    """out = s * (S act(S x Wd^T + bd) Wu^T + bu + [skip] x) on a [N, d] input.

    The host side of a call is kept short on purpose (molecule batches are launch-bound):
    one allocation for the saved rank-width tensors, one for the workspace, sizes cached on
    the plan, no device-guard context when the tensor already lives on the current device."""

    @staticmethod
    def forward(ctx, x, w_down, b_down, w_up, b_up, scalar, plan: GraphPlan, act: int, skip: bool, walk: int = 0):
        lib = _lib.load()
        if not x.is_cuda or x.dtype != torch.float32:
            _require_hip_f32(x, "x")
        n, d = x.shape
        r = w_down.shape[0]
        if n != plan.num_nodes:
            raise RuntimeError(f"x has {n} nodes but the graph plan was built for {plan.num_nodes}")
        fp, ws_fwd, _ = plan.sizes(lib, d, r)
        if not x.is_contiguous():
            x = x.contiguous()
        if not w_down.is_contiguous():
            w_down = w_down.contiguous()
        if not w_up.is_contiguous():
            w_up = w_up.contiguous()
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        out = torch.empty_like(x)
        saved = torch.empty(3 if act == 2 else 2, n, fp, dtype=torch.float32, device=dev)
        saved_h, saved_p = saved[0], saved[1]
        saved_h1 = saved[2] if act == 2 else None
        ws = torch.empty(ws_fwd, dtype=torch.uint8, device=dev)
        params = _lib.AdapterParams(w_down.data_ptr(), _ptr(b_down), w_up.data_ptr(), _ptr(b_up),
                                    _ptr(scalar), d, r, act, int(skip), int(walk))
        st = lib.gconv_adapter_fwd_f32(plan.ref(), C.byref(params), x.data_ptr(), out.data_ptr(),
                                       saved_h.data_ptr(), _ptr(saved_h1), saved_p.data_ptr(),
                                       ws.data_ptr(), ws_fwd, _stream(dev.index))
        if st != 0:
            _lib.check(st, "adapter_fwd")
        ctx.plan, ctx.act, ctx.skip, ctx.walk = plan, act, skip, int(walk)
        ctx.save_for_backward(x, w_down, b_down, w_up, b_up, scalar, saved)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x, w_down, b_down, w_up, b_up, scalar, saved = ctx.saved_tensors
        plan: GraphPlan = ctx.plan
        g = grad_out if grad_out.is_contiguous() else grad_out.contiguous()
        if not g.is_cuda or g.dtype != torch.float32:
            _require_hip_f32(g, "grad_output")
        n, d = x.shape
        r = w_down.shape[0]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        _, _, ws_bwd = plan.sizes(lib, d, r)
        act = ctx.act
        saved_h, saved_p = saved[0], saved[1]
        saved_h1 = saved[2] if act == 2 else None
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gwd, gwu = torch.empty_like(w_down), torch.empty_like(w_up)
        gbd = torch.empty_like(b_down) if b_down is not None else None
        gbu = torch.empty_like(b_up) if b_up is not None else None
        gs = torch.empty_like(scalar) if scalar is not None else None
        ws = torch.empty(ws_bwd, dtype=torch.uint8, device=dev)
        params = _lib.AdapterParams(w_down.data_ptr(), _ptr(b_down), w_up.data_ptr(), _ptr(b_up),
                                    _ptr(scalar), d, r, act, int(ctx.skip), ctx.walk)
        st = lib.gconv_adapter_bwd_f32(plan.ref(), C.byref(params), x.data_ptr(), g.data_ptr(),
                                       saved_h.data_ptr(), _ptr(saved_h1), saved_p.data_ptr(),
                                       _ptr(gx), gwd.data_ptr(), _ptr(gbd), gwu.data_ptr(),
                                       _ptr(gbu), _ptr(gs), ws.data_ptr(), ws_bwd, _stream(dev.index))
        if st != 0:
            _lib.check(st, "adapter_bwd")
        return gx, gwd, gbd, gwu, gbu, gs, None, None, None, None


def gconv_adapter(x: torch.Tensor, plan: GraphPlan, w_down, b_down, w_up, b_up,
                  scalar: Optional[torch.Tensor], act: str = "relu", skip: bool = True) -> torch.Tensor:
    """Functional form on a 2-D ``[N, d]`` input."""
    return GConvAdapterFunction.apply(x, w_down, b_down, w_up, b_up, scalar, plan,
                                      _lib.ACT_CODES[act], bool(skip))


# ---------------------------------------------------------------------------
# stand-alone kernels (no autograd): parity tests and roofline micro-benchmarks
# ---------------------------------------------------------------------------
def spmm(plan: GraphPlan, h: torch.Tensor, *, transpose: bool = False,
         bias: Optional[torch.Tensor] = None, act: str = "none", flags: int = 0,
         out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None
         ) -> torch.Tensor:
    """Y = act(S h + bias) (or S^T with ``transpose``); h is [N, F], F in {4, 8, 16, 32, 64}."""
    lib = _lib.load()
    _require_hip_f32(h, "h")
    h = h.contiguous()
    n, f = h.shape
    csr = plan.bwd if transpose else plan.fwd
    y = out if out is not None else torch.empty_like(h)
    with torch.cuda.device(h.device):
        if workspace is None:
            ws_bytes = int(lib.gconv_spmm_workspace_bytes(csr.ref(), f))
            workspace = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=h.device)
        st = lib.gconv_spmm_csr_f32(csr.ref(), plan.dis_ptr, h.data_ptr(), y.data_ptr(), f,
                                    _lib.ptr(bias), _lib.ACT_CODES[act], flags,
                                    workspace.data_ptr(), workspace.numel(), _stream())
    _lib.check(st, "spmm_csr_f32")
    return y


def gemm_down(x: torch.Tensor, w_down: torch.Tensor, row_scale: Optional[torch.Tensor] = None
              ) -> torch.Tensor:
    """[N, Fp] = row_scale .* (x @ w_down.T), zero-padded to the padded rank."""
    lib = _lib.load()
    _require_hip_f32(x, "x")
    x, w_down = x.contiguous(), w_down.contiguous()
    n, d = x.shape
    r = w_down.shape[0]
    fp = int(lib.gconv_padded_rank(r))
    if fp == 0:
        raise RuntimeError(f"bottleneck_size={r} is unsupported (1..64)")
    h0 = torch.empty(n, fp, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.gconv_gemm_down_f32(x.data_ptr(), w_down.data_ptr(), _lib.ptr(row_scale),
                                     h0.data_ptr(), n, d, r, _stream())
    _lib.check(st, "gemm_down_f32")
    return h0


def gemm_up_fused(p: torch.Tensor, w_up: torch.Tensor, b_up: Optional[torch.Tensor],
                  scalar: Optional[torch.Tensor], x: Optional[torch.Tensor], skip: bool = True
                  ) -> torch.Tensor:
    """[N, d] = s * (p[:, :r] @ w_up.T + b_up + x)."""
    lib = _lib.load()
    _require_hip_f32(p, "p")
    p, w_up = p.contiguous(), w_up.contiguous()
    n = p.shape[0]
    d, r = w_up.shape
    out = torch.empty(n, d, dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        st = lib.gconv_gemm_up_fused_f32(p.data_ptr(), w_up.data_ptr(), _lib.ptr(b_up),
                                         _lib.ptr(scalar), _lib.ptr(x), int(skip),
                                         out.data_ptr(), n, d, r, _stream())
    _lib.check(st, "gemm_up_fused_f32")
    return out


class LayerNormScaleFunction(torch.autograd.Function):
    """out = scalar * LayerNorm(z) — the adapter's ``layer_norm`` tail
    (``src/finetune/gconv_adapter.py:101-102,105-106``) in one HIP pass each way
    (``csrc/norm.hip``) instead of torch's LayerNorm + multiply and their five backward kernels."""

    @staticmethod
    def forward(ctx, z, gamma, beta, scalar, eps: float):
        lib = _lib.load()
        _require_hip_f32(z, "z")
        shape = z.shape
        z2 = z.contiguous().view(-1, shape[-1])
        n, d = z2.shape
        dev = z.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        out = torch.empty_like(z2)
        stats = torch.empty(n, 2, dtype=torch.float32, device=dev)
        st = lib.gconv_layer_norm_scale_fwd_f32(z2.data_ptr(), _ptr(gamma), _ptr(beta), _ptr(scalar),
                                                out.data_ptr(), stats.data_ptr(), n, d, float(eps), _stream(dev.index))
        if st != 0:
            _lib.check(st, "layer_norm_scale_fwd")
        ctx.save_for_backward(z2, stats, gamma, beta, scalar)
        ctx.shape = shape
        return out.view(shape)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        z2, stats, gamma, beta, scalar = ctx.saved_tensors
        n, d = z2.shape
        g2 = g.contiguous().view(n, d)
        dev = z2.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        gz = torch.empty_like(z2) if ctx.needs_input_grad[0] else None
        ggam = torch.empty_like(gamma) if gamma is not None else None
        gbet = torch.empty_like(beta) if beta is not None else None
        gsc = torch.empty_like(scalar) if scalar is not None else None
        nbytes = int(lib.gconv_layer_norm_scale_bwd_workspace_bytes(n, d))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        st = lib.gconv_layer_norm_scale_bwd_f32(g2.data_ptr(), z2.data_ptr(), stats.data_ptr(), _ptr(gamma),
                                                _ptr(beta), _ptr(scalar), _ptr(gz), _ptr(ggam), _ptr(gbet),
                                                _ptr(gsc), n, d, ws.data_ptr(), ws.numel(), _stream(dev.index))
        if st != 0:
            _lib.check(st, "layer_norm_scale_bwd")
        return (gz.view(ctx.shape) if gz is not None else None), ggam, gbet, gsc, None
