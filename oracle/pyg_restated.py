"""Edge-list restatement of the GConv-Adapter hot path in PyG's op order.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``) — PARITY UNPINNED: the
reference holds no golden vectors and ``torch-geometric`` cannot be imported
here; this file restates the published semantics of
``torch_geometric.nn.GCNConv`` (2.5.3 / 1.7.2) at the reference's call sites.

Reference anchors (``/root/reference``):
  * composition, flags, residual / norm / scalar order
        ``src/finetune/gconv_adapter.py:80-108``
  * construction of the two convs and the ``normalize`` kwarg
        ``src/finetune/gconv_adapter.py:40-41``
  * parameter names / shapes / init
        ``src/finetune/gconv_adapter.py:65,73-78``

Restated PyG semantics:
  1. ``edge_index[0]`` = source j, ``edge_index[1]`` = target i; sum on target.
  2. duplicate edges are not coalesced.
  3. ``normalize=False``: no self-loops added, unit weights.
  4. ``normalize=True``: ``add_remaining_self_loops`` (existing loops dropped,
     one loop of weight 1 per node), in-degree on target of the looped graph,
     ``dis = deg^-1/2`` (inf -> 0), ``w = dis[src] * dis[dst]``; recomputed on
     every conv call (``cached=False``).
  5. bias-free linear first, then propagate, then ``+ bias``.
  6. node dimension is ``-2`` (``x`` may be ``[N, F]`` or ``[B, N, F]``).

Everything is plain torch on CPU tensors; dtype follows ``x`` (fp32 for the
timed CPU baseline, fp64 for golden vectors).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# gcn_norm  (PyG: torch_geometric/nn/conv/gcn_conv.py :: gcn_norm, Tensor path)
# ----------------------------------------------------------------------------
def add_remaining_self_loops(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Drop existing self-loops, append exactly one loop per node (weight 1)."""
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index[:, keep], loops.unsqueeze(0).repeat(2, 1)], dim=1)


def gcn_norm(edge_index: torch.Tensor, num_nodes: int, dtype: torch.dtype
             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (looped edge_index [2, E~], symmetric-normalised weights [E~])."""
    ei = add_remaining_self_loops(edge_index, num_nodes)
    w = torch.ones(ei.size(1), dtype=dtype, device=ei.device)
    src, dst = ei[0], ei[1]
    deg = torch.zeros(num_nodes, dtype=dtype, device=ei.device).index_add_(0, dst, w)
    dis = deg.pow(-0.5)
    dis = dis.masked_fill(dis == float("inf"), 0.0)
    return ei, dis[src] * w * dis[dst]


# ----------------------------------------------------------------------------
# GCNConv.forward  (lin -> propagate(add) -> + bias)
# ----------------------------------------------------------------------------
def propagate(h: torch.Tensor, edge_index: torch.Tensor, weight: Optional[torch.Tensor],
              num_nodes: int) -> torch.Tensor:
    """``MessagePassing.propagate`` with aggr='add', flow source->target, node_dim=-2."""
    src, dst = edge_index[0], edge_index[1]
    msg = h.index_select(-2, src)
    if weight is not None:
        msg = weight.view(-1, 1) * msg
    out_shape = list(h.shape)
    out_shape[-2] = num_nodes
    out = torch.zeros(out_shape, dtype=h.dtype, device=h.device)
    return out.index_add_(h.dim() - 2, dst, msg)


def gcn_conv(x: torch.Tensor, edge_index: torch.Tensor, lin_weight: torch.Tensor,
             bias: Optional[torch.Tensor], normalize: bool) -> torch.Tensor:
    num_nodes = x.size(-2)
    weight = None
    if normalize:  # recomputed per call, as PyG does with cached=False
        edge_index, weight = gcn_norm(edge_index, num_nodes, x.dtype)
    h = F.linear(x, lin_weight)
    out = propagate(h, edge_index, weight, num_nodes)
    if bias is not None:
        out = out + bias
    return out


# ----------------------------------------------------------------------------
# GConvAdapter.forward  (src/finetune/gconv_adapter.py:80-108)
# ----------------------------------------------------------------------------
@dataclass
class AdapterFlags:
    non_linearity: str = "relu"        # relu | silu | none      (:44-51)
    normalization: str = "none"        # batch_norm | layer_norm | none (:54-61)
    learnable_scalar: bool = False     # (:64-67)
    skip_connection: bool = True       # (:70)
    normalize: bool = True             # (:40-41)


def _act(name: str, h: torch.Tensor) -> torch.Tensor:
    if name == "relu":
        return F.relu(h)
    if name == "silu":
        return F.silu(h)
    if name == "none":
        return h
    raise ValueError("Invalid non_linearity. Supported types: 'relu', 'silu', 'none'.")


BN_EPS, BN_MOMENTUM = 1e-5, 0.1   # nn.BatchNorm1d defaults (the reference constructs it with none, :54-55)


def adapter_forward(x: torch.Tensor, edge_index: torch.Tensor, params: Dict[str, torch.Tensor],
                    flags: AdapterFlags, training: bool = True) -> torch.Tensor:
    """``params`` keys follow the reference state_dict: conv_down.lin.weight [r,d],
    conv_down.bias [r], conv_up.lin.weight [d,r], conv_up.bias [d], scalar [1],
    normalization.weight / normalization.bias [d] (optional) and, for ``batch_norm``, the module's
    buffers normalization.running_mean / normalization.running_var [d] (optional).

    ``training`` is ``module.training`` of the reference module: the train loop calls ``model.train()``
    (``scripts/finetune_inductive_learning.py:38``), the three evaluations per epoch ``model.eval()`` under
    ``torch.no_grad()`` (``:95-104``). Only the ``batch_norm`` tail depends on it, exactly as
    ``nn.BatchNorm1d.forward`` does (track_running_stats=True, momentum 0.1, eps 1e-5):
      training  batch statistics (biased variance) normalise; the running buffers, when given, are
                updated IN PLACE with momentum 0.1 and the UNBIASED batch variance;
      eval      the running buffers normalise (no buffers: nothing to evaluate with -> ValueError, where
                torch would fall back to batch statistics only for track_running_stats=False)."""
    h = gcn_conv(x, edge_index, params["conv_down.lin.weight"], params.get("conv_down.bias"),
                 flags.normalize)
    h = _act(flags.non_linearity, h)
    out = gcn_conv(h, edge_index, params["conv_up.lin.weight"], params.get("conv_up.bias"),
                   flags.normalize)
    if flags.skip_connection:
        out = out + x                                     # :94-95 (in-place there)
    if flags.normalization == "batch_norm":
        if out.size(0) > 1:                               # :99
            rm, rv = params.get("normalization.running_mean"), params.get("normalization.running_var")
            if not training and (rm is None or rv is None):
                raise ValueError("adapter_forward(training=False) with batch_norm needs normalization.running_mean / "
                                 "normalization.running_var (the module's buffers)")
            out = F.batch_norm(out, rm, rv, params.get("normalization.weight"), params.get("normalization.bias"),
                               training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    elif flags.normalization == "layer_norm":
        d = out.size(-1)
        out = F.layer_norm(out, (d,), params.get("normalization.weight"),
                           params.get("normalization.bias"))
    elif flags.normalization != "none":
        raise ValueError("Invalid normalization. Supported types: 'batch_norm', 'layer_norm', 'none'.")
    if flags.learnable_scalar:
        out = out * params["scalar"]                      # :105-106
    return out


def adapter_forward_backward(x: torch.Tensor, edge_index: torch.Tensor,
                             params: Dict[str, torch.Tensor], flags: AdapterFlags,
                             grad_out: torch.Tensor, need_dx: bool = True
                             ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Forward + autograd backward with upstream gradient ``grad_out``.

    Returns (out, grads) where grads has one entry per parameter plus ``x``."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()
              if not k.startswith("normalization.running_")}          # (buffers are not leaves)
    xl = x.detach().clone().requires_grad_(need_dx)
    out = adapter_forward(xl, edge_index, leaves, flags)
    out.backward(grad_out)
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    if need_dx:
        grads["x"] = xl.grad
    return out.detach(), grads


# ----------------------------------------------------------------------------
# Reassociated form (both propagations at rank r) — what the HIP path computes
# ----------------------------------------------------------------------------
def adapter_forward_reassociated(x: torch.Tensor, edge_index: torch.Tensor,
                                 params: Dict[str, torch.Tensor], flags: AdapterFlags
                                 ) -> torch.Tensor:
    """S(H W^T) == (S H) W^T: same math as :func:`adapter_forward` with the second
    propagation done at rank r (no normalization layer; used for timing and
    for the fp32 reassociation-error budget)."""
    n = x.size(-2)
    ei, w = (gcn_norm(edge_index, n, x.dtype) if flags.normalize else (edge_index, None))
    h0 = F.linear(x, params["conv_down.lin.weight"])
    h1 = propagate(h0, ei, w, n) + params["conv_down.bias"]
    h2 = _act(flags.non_linearity, h1)
    p = propagate(h2, ei, w, n)
    out = F.linear(p, params["conv_up.lin.weight"]) + params["conv_up.bias"]
    if flags.skip_connection:
        out = out + x
    if flags.learnable_scalar:
        out = out * params["scalar"]
    return out


def init_params(hidden_size: int, bottleneck_size: int, flags: AdapterFlags,
                std: float = 0.1, seed: int = 0, dtype: torch.dtype = torch.float64
                ) -> Dict[str, torch.Tensor]:
    """Trained-scale parameters (std 0.1, non-zero biases): at the reference's
    init scale (std 1e-5, ``gconv_adapter.py:73-78``) the adapter is a
    near-identity and a parity test would be vacuous."""
    g = torch.Generator().manual_seed(seed)
    p = {
        "conv_down.lin.weight": torch.randn(bottleneck_size, hidden_size, generator=g, dtype=dtype) * std,
        "conv_down.bias": torch.randn(bottleneck_size, generator=g, dtype=dtype) * std,
        "conv_up.lin.weight": torch.randn(hidden_size, bottleneck_size, generator=g, dtype=dtype) * std,
        "conv_up.bias": torch.randn(hidden_size, generator=g, dtype=dtype) * std,
    }
    if flags.learnable_scalar:
        p["scalar"] = torch.tensor([0.75], dtype=dtype)
    if flags.normalization in ("batch_norm", "layer_norm"):
        p["normalization.weight"] = 1.0 + 0.1 * torch.randn(hidden_size, generator=g, dtype=dtype)
        p["normalization.bias"] = 0.1 * torch.randn(hidden_size, generator=g, dtype=dtype)
    return p
