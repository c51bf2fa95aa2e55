"""NodeFormer's relational bias on the adapter's SpMM kernel.

``add_conv_relational_bias(x, edge_index, b, trans)`` of the reference
(``/root/reference/src/models/transductive/nodeformer.py:202-224``, called once per relational-bias
order and layer, ``:286-287``) builds one ``torch_sparse.SparseTensor`` per head and multiplies.
It is ``out = trans(b_i) * D_in^-1/2 A_in D_out^-1/2 x`` — a row scaling, ONE unnormalised CSR
product over the graph plan the adapters of the same model already share (duplicates and loops
count exactly as given: the plan with ``normalize=False`` drops nothing), a row scaling. All heads
go through the same product (a head only differs by its scalar): ``[N, H * D]`` at once when that
width is one the SpMM has (4..64), per head otherwise."""
from __future__ import annotations

import torch

from . import _lib
from .functional import spmm
from .graph import GraphPlan, get_plan

_WIDTHS = (4, 8, 16, 32, 64)


class SpmmFunction(torch.autograd.Function):
    """Y = A_in H on a graph plan (unit weights); d H = A_in^T d Y."""

    @staticmethod
    def forward(ctx, h: torch.Tensor, plan: GraphPlan) -> torch.Tensor:
        ctx.plan = plan
        return spmm(plan, h)

    @staticmethod
    def backward(ctx, g):
        return spmm(ctx.plan, g.contiguous(), transpose=True), None


def _inv_sqrt_degrees(plan: GraphPlan):
    """(d_in^-1/2, d_out^-1/2) as fp32 [N, 1] (0 where the degree is 0), cached on the plan."""
    hit = plan.__dict__.get("_rb_scales")
    if hit is None:
        def inv_sqrt(rowptr):
            d = (rowptr[1:] - rowptr[:-1]).to(torch.float32)
            return torch.where(d > 0, d.rsqrt(), torch.zeros_like(d)).unsqueeze(1)
        s_in = inv_sqrt(plan.fwd.rowptr)
        s_out = s_in if plan.symmetric else inv_sqrt(plan.bwd.rowptr)
        hit = plan.__dict__["_rb_scales"] = (s_in, s_out)
    return hit


def add_conv_relational_bias(x: torch.Tensor, edge_index: torch.Tensor, b: torch.Tensor,
                             trans: str = "sigmoid") -> torch.Tensor:
    """Same signature and result as the reference function: ``x [B, N, H, D]`` (the reference passes
    B = 1), ``edge_index [2, E]`` (row = source, col = target), ``b [H]``."""
    if trans == "sigmoid":
        scale = torch.sigmoid(b)
    elif trans == "identity":
        scale = b
    else:
        raise NotImplementedError
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError("gconv_adapter_amd.add_conv_relational_bias takes an fp32 [B, N, H, D] tensor on a HIP "
                           "device (no CPU path)")
    bsz, n, heads, d = x.shape
    plan = get_plan(edge_index, n, False)
    s_in, s_out = _inv_sqrt_degrees(plan)
    outs = []
    for bi in range(bsz):
        xb = x[bi].reshape(n, heads * d)
        if heads * d in _WIDTHS:
            y = SpmmFunction.apply((xb * s_out).contiguous(), plan) * s_in
            y = y.view(n, heads, d) * scale.view(1, heads, 1)
        elif d in _WIDTHS:
            y = torch.stack([SpmmFunction.apply((x[bi, :, i] * s_out).contiguous(), plan) * (s_in * scale[i])
                             for i in range(heads)], dim=1)
        else:
            raise RuntimeError(f"head width {d} is unsupported (the SpMM has widths {_WIDTHS})")
        outs.append(y)
    return torch.stack(outs, dim=0)
