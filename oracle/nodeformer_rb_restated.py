"""Restatement of NodeFormer's relational bias — TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``/root/reference/src/models/transductive/nodeformer.py:202-224``
(``add_conv_relational_bias``): per head ``i``

    value_e = trans(b[i]) * d_in[col_e]^-1/2 * d_out[row_e]^-1/2
    adj_i   = SparseTensor(row=col, col=row, value)          # entry (col_e, row_e)
    out[:, :, i] = adj_i @ x[:, :, i]                        # out[col_e] += value_e * x[row_e]

with ``row, col = edge_index``, ``d_in = degree(col)``, ``d_out = degree(row)`` (all entries count:
self-loops and duplicates included). ``torch_sparse`` is not installed here; the product is written
with ``index_add_`` (parity unpinned, as for the adapter itself)."""
from __future__ import annotations

import torch


def add_conv_relational_bias(x: torch.Tensor, edge_index: torch.Tensor, b: torch.Tensor,
                             trans: str = "sigmoid") -> torch.Tensor:
    """x ``[B, N, H, D]``, edge_index ``[2, E]``, b ``[H]`` -> ``[B, N, H, D]``."""
    row, col = edge_index[0], edge_index[1]
    n = x.shape[1]
    ones = torch.ones(row.shape[0], dtype=x.dtype)
    d_in = torch.zeros(n, dtype=x.dtype).index_add_(0, col, ones)
    d_out = torch.zeros(n, dtype=x.dtype).index_add_(0, row, ones)
    d_norm_in = (1.0 / d_in[col]).sqrt()
    d_norm_out = (1.0 / d_out[row]).sqrt()
    outs = []
    for i in range(x.shape[2]):
        if trans == "sigmoid":
            b_i = torch.sigmoid(b[i])
        elif trans == "identity":
            b_i = b[i]
        else:
            raise NotImplementedError
        value = b_i * d_norm_in * d_norm_out                                   # [E]
        xi = x[:, :, i]                                                        # [B, N, D]
        msg = value.view(1, -1, 1) * xi.index_select(1, row)                   # entry (col_e, row_e) * x[row_e]
        outs.append(torch.zeros_like(xi).index_add_(1, col, msg))
    return torch.stack(outs, dim=2)
