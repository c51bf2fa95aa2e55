"""Edge-list restatement of the frozen GIN layer next to the adapter.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``) — PARITY UNPINNED for the same reason
as the adapter: ``MolecularGINConv`` is a ``torch_geometric.nn.MessagePassing`` subclass, PyG
cannot be imported here and the reference holds no fixtures for it. The layer's own code is in
the reference, so the restatement follows it line by line
(``/root/reference/src/layers/inductive/gin_conv.py``) and only ``propagate`` (gather source
rows, sum on target) and ``add_self_loops`` (append one (i, i) edge per node, existing loops
kept) are restated from PyG's published semantics:

  * ``:50``     ``edge_index = add_self_loops(edge_index, num_nodes)``
  * ``:53-55``  self-loop attributes: bond type 4, direction 0, appended after the real edges
  * ``:57``     ``edge_embeddings = type_table[attr[:, 0]] + direction_table[attr[:, 1]]``
  * ``:59,61-72`` message ``x_j + edge_attr``, aggregated with ``aggr="add"`` on the target
  * ``:74-84``  ``update``: ``Linear(d, 2d) -> ReLU -> Linear(2d, d)`` (``:24-28``)

Plain torch on CPU tensors, PyG's op order (index_select -> add -> index_add_); dtype follows
``x`` (fp64 for the parity checks).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

NUM_BOND_TYPE = 6        # gin_conv.py:6
NUM_BOND_DIRECTION = 3   # gin_conv.py:7
SELF_LOOP_BOND_TYPE = 4  # gin_conv.py:54


def init_params(d: int, seed: int = 0, dtype=torch.float64) -> Dict[str, torch.Tensor]:
    """Random parameters under the reference's state-dict names (``gin_conv.py:24-32``)."""
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s, std=0.1: (std * torch.randn(*s, generator=g, dtype=torch.float64)).to(dtype)
    return {"mlp.0.weight": mk(2 * d, d), "mlp.0.bias": mk(2 * d), "mlp.2.weight": mk(d, 2 * d),
            "mlp.2.bias": mk(d), "edge_embedding_type.weight": mk(NUM_BOND_TYPE, d, std=0.5),
            "edge_embedding_direction.weight": mk(NUM_BOND_DIRECTION, d, std=0.5)}


def gin_aggregate(x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor,
                  params: Dict[str, torch.Tensor]) -> torch.Tensor:
    """``forward`` up to ``update`` (``gin_conv.py:38-73``)."""
    n = x.shape[0]
    loops = torch.arange(n, dtype=edge_index.dtype)
    ei = torch.cat([edge_index, torch.stack([loops, loops])], dim=1)                  # :50
    self_attr = torch.zeros(n, 2, dtype=edge_attr.dtype)
    self_attr[:, 0] = SELF_LOOP_BOND_TYPE                                              # :53-54
    attr = torch.cat([edge_attr.reshape(-1, 2), self_attr], dim=0)                    # :55
    emb = params["edge_embedding_type.weight"][attr[:, 0]] + \
        params["edge_embedding_direction.weight"][attr[:, 1]]                         # :57
    msg = x.index_select(0, ei[0]) + emb                                               # :61-72
    return torch.zeros_like(x).index_add_(0, ei[1], msg)                               # aggr="add"


def gin_forward(x, edge_index, edge_attr, params):
    """The whole layer (``gin_conv.py:38-84``)."""
    h = gin_aggregate(x, edge_index, edge_attr, params)
    h = F.relu(F.linear(h, params["mlp.0.weight"], params["mlp.0.bias"]))
    return F.linear(h, params["mlp.2.weight"], params["mlp.2.bias"])


def gin_forward_backward(x, edge_index, edge_attr, params, grad_out):
    """out and the gradients w.r.t. x and every parameter (autograd on the restatement)."""
    x = x.clone().requires_grad_(True)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = gin_forward(x, edge_index, edge_attr, p)
    out.backward(grad_out)
    grads = {"x": x.grad.detach()}
    grads.update({k: v.grad.detach() for k, v in p.items()})
    return out.detach(), grads


def gin_aggregate_dense(x, edge_index, edge_attr, params):
    """Second derivation for small graphs: dense (A + I) x plus the per-node bond sums."""
    n = x.shape[0]
    a = torch.zeros(n, n, dtype=x.dtype)
    a.index_put_((edge_index[1], edge_index[0]), torch.ones(edge_index.shape[1], dtype=x.dtype), accumulate=True)
    out = (a + torch.eye(n, dtype=x.dtype)) @ x
    t, d_ = params["edge_embedding_type.weight"], params["edge_embedding_direction.weight"]
    out = out + t[SELF_LOOP_BOND_TYPE] + d_[0]
    ea = edge_attr.reshape(-1, 2)
    for e in range(edge_index.shape[1]):
        out[edge_index[1, e]] += t[ea[e, 0]] + d_[ea[e, 1]]
    return out
