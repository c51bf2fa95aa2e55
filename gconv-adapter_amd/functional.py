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


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _params_struct(w_down, b_down, w_up, b_up, scalar, act: int, skip: bool) -> _lib.AdapterParams:
    r, d = w_down.shape
    return _lib.AdapterParams(w_down.data_ptr(), _lib.ptr(b_down), w_up.data_ptr(), _lib.ptr(b_up),
                              _lib.ptr(scalar), d, r, act, int(skip))


class GConvAdapterFunction(torch.autograd.Function):
    """out = s * (S act(S x Wd^T + bd) Wu^T + bu + [skip] x) on a [N, d] input."""

    @staticmethod
    def forward(ctx, x, w_down, b_down, w_up, b_up, scalar, plan: GraphPlan, act: int, skip: bool):
        lib = _lib.load()
        _require_hip_f32(x, "x")
        n, d = x.shape
        r = w_down.shape[0]
        if n != plan.num_nodes:
            raise RuntimeError(f"x has {n} nodes but the graph plan was built for {plan.num_nodes}")
        fp = int(lib.gconv_padded_rank(r))
        if fp == 0:
            raise RuntimeError(f"bottleneck_size={r} is unsupported (1..64)")
        x = x.contiguous()
        w_down, w_up = w_down.contiguous(), w_up.contiguous()
        out = torch.empty_like(x)
        saved_h = torch.empty(n, fp, dtype=torch.float32, device=x.device)
        saved_p = torch.empty(n, fp, dtype=torch.float32, device=x.device)
        saved_h1 = torch.empty(n, fp, dtype=torch.float32, device=x.device) if act == 2 else None
        params = _params_struct(w_down, b_down, w_up, b_up, scalar, act, skip)
        with torch.cuda.device(x.device):
            ws_bytes = int(lib.gconv_adapter_fwd_workspace_bytes(plan.ref(), d, r))
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
            st = lib.gconv_adapter_fwd_f32(plan.ref(), C.byref(params),
                                           x.data_ptr(), out.data_ptr(), saved_h.data_ptr(),
                                           _lib.ptr(saved_h1), saved_p.data_ptr(), ws.data_ptr(),
                                           ws.numel(), _stream())
        _lib.check(st, "adapter_fwd")
        ctx.plan, ctx.act, ctx.skip = plan, act, skip
        ctx.has = (b_down is not None, b_up is not None, scalar is not None, saved_h1 is not None)
        ctx.save_for_backward(x, w_down, b_down, w_up, b_up, scalar, saved_h, saved_h1, saved_p)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x, w_down, b_down, w_up, b_up, scalar, saved_h, saved_h1, saved_p = ctx.saved_tensors
        plan: GraphPlan = ctx.plan
        g = grad_out.contiguous()
        _require_hip_f32(g, "grad_output")
        n, d = x.shape
        r = w_down.shape[0]
        dev = x.device
        need_dx = ctx.needs_input_grad[0]
        gx = torch.empty_like(x) if need_dx else None
        gwd, gwu = torch.empty_like(w_down), torch.empty_like(w_up)
        gbd = torch.empty_like(b_down) if b_down is not None else None
        gbu = torch.empty_like(b_up) if b_up is not None else None
        gs = torch.empty_like(scalar) if scalar is not None else None
        params = _params_struct(w_down, b_down, w_up, b_up, scalar, ctx.act, ctx.skip)
        with torch.cuda.device(dev):
            ws_bytes = int(lib.gconv_adapter_bwd_workspace_bytes(plan.ref(), d, r))
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
            st = lib.gconv_adapter_bwd_f32(plan.ref(), C.byref(params),
                                           x.data_ptr(), g.data_ptr(), saved_h.data_ptr(),
                                           _lib.ptr(saved_h1), saved_p.data_ptr(), _lib.ptr(gx),
                                           gwd.data_ptr(), _lib.ptr(gbd), gwu.data_ptr(),
                                           _lib.ptr(gbu), _lib.ptr(gs), ws.data_ptr(), ws.numel(),
                                           _stream())
        _lib.check(st, "adapter_bwd")
        return gx, gwd, gbd, gwu, gbu, gs, None, None, None


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
