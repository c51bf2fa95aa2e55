"""Shared test helpers: golden loading and oracle <-> module plumbing."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch

from oracle import pyg_restated as oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAM_KEYS = ["conv_down.lin.weight", "conv_down.bias", "conv_up.lin.weight", "conv_up.bias",
              "scalar", "normalization.weight", "normalization.bias"]
RTOL = ATOL = 1e-5   # north_star tolerance, vs the fp64 oracle


def flags_from_record(rec) -> oracle.AdapterFlags:
    act, norm, scalar, skip, normalize = [str(v) for v in rec]
    return oracle.AdapterFlags(non_linearity=act, normalization=norm,
                               learnable_scalar=scalar == "True", skip_connection=skip == "True",
                               normalize=normalize == "True")


def load_golden(name: str) -> dict:
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def golden_case(rec: dict, index=None) -> dict:
    """Splits a golden record into flags / inputs / params / expected (torch tensors)."""
    def get(k):
        v = rec[k]
        return v[index] if index is not None else v
    flags = flags_from_record(get("flags"))
    params = {}
    for k in PARAM_KEYS:
        if "param." + k not in rec:
            continue
        if k == "scalar" and not flags.learnable_scalar:
            continue
        if k.startswith("normalization") and flags.normalization == "none":
            continue
        params[k] = torch.from_numpy(np.asarray(get("param." + k)))
    expected = {"out": torch.from_numpy(np.asarray(get("out"))),
                "x": torch.from_numpy(np.asarray(get("grad.x")))}
    for k in params:
        expected[k] = torch.from_numpy(np.asarray(get("grad." + k)))
    return dict(flags=flags, x=torch.from_numpy(np.asarray(get("x"))),
                edge_index=torch.from_numpy(np.asarray(get("edge_index"))),
                grad_out=torch.from_numpy(np.asarray(get("grad_out"))),
                params=params, expected=expected)


def build_module(ga, d: int, r: int, flags: oracle.AdapterFlags, params: Dict[str, torch.Tensor], device):
    mod = ga.GConvAdapter(d, r, non_linearity=flags.non_linearity, normalization=flags.normalization,
                          learnable_scalar=flags.learnable_scalar,
                          skip_connection=flags.skip_connection, normalize=flags.normalize).to(device)
    with torch.no_grad():
        for k, v in params.items():
            mod.get_parameter(k).copy_(v.float())
    mod.train()
    return mod


def run_module(mod, x: torch.Tensor, edge_index: torch.Tensor, grad_out: torch.Tensor, device,
               need_dx: bool = True):
    xd = x.to(device).requires_grad_(need_dx)
    out = mod(xd, edge_index.to(device))
    mod.zero_grad(set_to_none=True)
    out.backward(grad_out.to(device))
    got = {"out": out.detach().cpu().double()}
    if need_dx:
        got["x"] = xd.grad.detach().cpu().double()
    for k, p in mod.named_parameters():
        got[k] = p.grad.detach().cpu().double()
    return got


def grad_atol(want: torch.Tensor, base: float = ATOL) -> float:
    """Absolute tolerance for PARAMETER gradients. dW / db / ds are sums over all N
    nodes with cancellation: an entry can be ~0 while the terms summed are O(max|dW|),
    and fp32 accumulation error scales with the terms, not with the entry. So the
    absolute term is taken relative to the tensor's scale: 1e-5 * max(1, max|want|).
    `out` and `dX` (no reduction over nodes) keep the plain element-wise 1e-5."""
    return base * max(1.0, float(want.abs().max())) if want.numel() else base


def assert_close(got: torch.Tensor, want: torch.Tensor, what: str, rtol=RTOL, atol=ATOL, row_scaled=False):
    """|got - want| <= atol + rtol |want| element-wise. `row_scaled`: the absolute term of a row is
    atol * max(1, max |want| over that row) — for UNNORMALISED sums over hub neighbourhoods, where an
    output row is a rank-r dot product of terms thousands of times larger than some of its entries
    (cancellation): fp32 rounding of the terms, in any summation order (the reference's own fp32
    path included), is then of the order 1e-7 x the row's scale, not of the entry."""
    want = want.double()
    err = (got - want).abs()
    scale = want.abs().amax(dim=-1, keepdim=True).clamp(min=1.0) if row_scaled and want.dim() >= 2 else 1.0
    tol = atol * scale + rtol * want.abs()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} != {tuple(want.shape)}"
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance, "
                           f"max abs err {err.max().item():.3e}, max |want| {want.abs().max().item():.3e}")


def check_workload(ga, dev, batch, d, r, flags, seed=0, need_dx=True, rtol=RTOL, atol=ATOL, row_scaled=False):
    """Module forward + backward on `dev` against the fp64 oracle (PyG op order, torch autograd) on
    the same seeded inputs: `out`, dX element-wise at (rtol, atol); parameter gradients with the
    scale-relative absolute term of `grad_atol`."""
    params = {k: v.float().double() for k, v in oracle.init_params(d, r, flags, seed=seed).items()}
    g = torch.randn(batch.x.shape, generator=torch.Generator().manual_seed(seed + 1))
    want_out, want_grads = oracle.adapter_forward_backward(batch.x.double(), batch.edge_index, params,
                                                           flags, g.double(), need_dx=need_dx)
    mod = build_module(ga, d, r, flags, params, dev)
    got = run_module(mod, batch.x, batch.edge_index, g, dev, need_dx=need_dx)
    assert_close(got["out"], want_out, "out", rtol, atol, row_scaled)
    for k, want in want_grads.items():
        assert_close(got[k], want, f"grad {k}", rtol, atol if k == "x" else grad_atol(want, atol),
                     row_scaled and k == "x")


class KinkWatch:
    """Counts, while the fp64 ORACLE model runs, the ReLU inputs that lie so close to zero that an fp32
    evaluation of the same model may take the other side of the kink (|pre| < band * max(1, rms(pre))
    of that activation tensor: an fp32 pre-activation is a sum of O(100) terms of that size, off by a few
    1e-7 of it). A unit that flips changes the gradient of every parameter upstream of
    it by O(1 / units), far above 1e-5 — so model-level tests look for seeded inputs on which the
    oracle has NO such unit (``risky == 0``) and then compare every gradient element-wise, instead of
    comparing gradient vectors by cosine. Patches ``torch.nn.functional.relu`` (what both oracle
    modules call) for the duration of the ``with`` block."""

    def __init__(self, band: float = 2e-5):
        self.band, self.risky, self.units = band, 0, 0

    def __enter__(self):
        import torch.nn.functional as F
        self._F, self._orig = F, F.relu

        def relu(x, inplace=False):
            a = x.detach().abs()
            if a.numel():
                self.risky += int((a < self.band * max(1.0, float(a.square().mean().sqrt()))).sum())
                self.units += a.numel()
            return self._orig(x, inplace)
        F.relu = relu
        return self

    def __exit__(self, *exc):
        self._F.relu = self._orig
        return False
