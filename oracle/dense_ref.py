"""Dense-matrix fp64 derivation of the GConv-Adapter forward and backward.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``) — PARITY UNPINNED.

Independent second derivation for small graphs: the propagation operator is
built as an explicit ``[N, N]`` float64 matrix ``S`` and the backward pass is
written out by hand instead of relying on autograd, so that
``oracle/pyg_restated.py`` (edge lists + autograd) and this file check each
other. numpy only.

Follows ``/root/reference/src/finetune/gconv_adapter.py:92-106`` for the
composition (conv_down -> act -> conv_up -> += x -> * scalar); normalization
layers are out of this file's scope (they are covered by the edge-list form).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def propagation_matrix(edge_index: np.ndarray, num_nodes: int, normalize: bool) -> np.ndarray:
    """S[i, j] = sum of weights of edges j -> i.

    normalize=False: every listed edge (duplicates and self-loops included) has
    weight 1.  normalize=True: existing self-loops are dropped, one loop per
    node is added, D~ = in-degree of the looped graph, S = D~^-1/2 (A0+I) D~^-1/2.
    """
    src = np.asarray(edge_index[0], dtype=np.int64)
    dst = np.asarray(edge_index[1], dtype=np.int64)
    a = np.zeros((num_nodes, num_nodes), dtype=np.float64)
    if normalize:
        keep = src != dst
        np.add.at(a, (dst[keep], src[keep]), 1.0)
        a += np.eye(num_nodes)
        deg = a.sum(axis=1)
        with np.errstate(divide="ignore"):
            dis = np.where(deg > 0, deg ** -0.5, 0.0)
        return dis[:, None] * a * dis[None, :]
    np.add.at(a, (dst, src), 1.0)
    return a


def _act(name: str, h: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (act(h), act'(h))."""
    if name == "relu":
        return np.maximum(h, 0.0), (h > 0).astype(h.dtype)
    if name == "silu":
        sig = 1.0 / (1.0 + np.exp(-h))
        return h * sig, sig * (1.0 + h * (1.0 - sig))
    if name == "none":
        return h, np.ones_like(h)
    raise ValueError("Invalid non_linearity. Supported types: 'relu', 'silu', 'none'.")


def forward_backward(x: np.ndarray, edge_index: np.ndarray, p: Dict[str, np.ndarray],
                     grad_out: np.ndarray, *, normalize: bool, non_linearity: str = "relu",
                     skip_connection: bool = True, learnable_scalar: bool = False
                     ) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """x [N, d]; returns (out, grads) in float64 using the hand-written backward."""
    x = np.asarray(x, dtype=np.float64)
    g = np.asarray(grad_out, dtype=np.float64)
    wd = np.asarray(p["conv_down.lin.weight"], dtype=np.float64)   # [r, d]
    bd = np.asarray(p["conv_down.bias"], dtype=np.float64)         # [r]
    wu = np.asarray(p["conv_up.lin.weight"], dtype=np.float64)     # [d, r]
    bu = np.asarray(p["conv_up.bias"], dtype=np.float64)           # [d]
    s = float(np.asarray(p["scalar"]).reshape(-1)[0]) if learnable_scalar else 1.0
    n = x.shape[0]
    S = propagation_matrix(edge_index, n, normalize)

    # forward (PyG order: linear, propagate, bias)
    h1 = S @ (x @ wd.T) + bd
    h2, dact = _act(non_linearity, h1)
    h3 = S @ (h2 @ wu.T) + bu
    pre = h3 + x if skip_connection else h3
    out = pre * s

    # backward
    grads: Dict[str, np.ndarray] = {}
    if learnable_scalar:
        grads["scalar"] = np.array([np.sum(g * pre)])
    g3 = s * g
    P = S @ h2
    grads["conv_up.bias"] = g3.sum(axis=0)
    grads["conv_up.lin.weight"] = g3.T @ P
    dP = g3 @ wu
    dh2 = S.T @ dP
    dh1 = dh2 * dact
    grads["conv_down.bias"] = dh1.sum(axis=0)
    T = S.T @ dh1
    grads["conv_down.lin.weight"] = T.T @ x
    dx = T @ wd
    if skip_connection:
        dx = dx + g3
    grads["x"] = dx
    return out, grads
