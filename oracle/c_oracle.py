"""ctypes loader of the plain-C oracle (``oracle/gconv_oracle.c``) — TEST INFRASTRUCTURE ONLY.

``build()`` compiles it with gcc into ``oracle/_build/libgconv_oracle.so`` (also called by
``__graft_entry__.build()``); ``adapter_forward_backward`` has the signature of
``pyg_restated.adapter_forward_backward`` for ``normalization='none'``."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Tuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "gconv_oracle.c")
LIB = os.path.join(HERE, "_build", "libgconv_oracle.so")
ACT = {"none": 0, "relu": 1, "silu": 2}
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.gconv_oracle_adapter_f64.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def adapter_forward_backward(x: torch.Tensor, edge_index: torch.Tensor, params: Dict[str, torch.Tensor],
                             flags, grad_out: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    if flags.normalization != "none":
        raise ValueError("the C oracle covers normalization='none'")
    lib = load()
    n, d = x.shape
    r = params["conv_down.lin.weight"].shape[0]
    f64 = lambda t: np.ascontiguousarray(t.detach().double().numpy()) if t is not None else None
    xs, go = f64(x), f64(grad_out)
    wd, wu = f64(params["conv_down.lin.weight"]), f64(params["conv_up.lin.weight"])
    bd, bu = f64(params.get("conv_down.bias")), f64(params.get("conv_up.bias"))
    sc = f64(params["scalar"]) if flags.learnable_scalar else None
    ei = np.ascontiguousarray(edge_index.numpy().astype(np.int64))
    out, gx = np.empty((n, d)), np.empty((n, d))
    gwd, gwu, gbd, gbu, gs = np.empty((r, d)), np.empty((d, r)), np.empty(r), np.empty(d), np.empty(1)
    rc = lib.gconv_oracle_adapter_f64(
        C.c_int64(n), C.c_int64(ei.shape[1]), _p(ei[0].copy()), _p(ei[1].copy()), d, r, _p(xs), _p(wd), _p(bd), _p(wu),
        _p(bu), _p(sc), ACT[flags.non_linearity], int(flags.skip_connection), int(flags.normalize), _p(go),
        _p(out), _p(gx), _p(gwd), _p(gbd), _p(gwu), _p(gbu), _p(gs))
    if rc != 0:
        raise RuntimeError(f"gconv_oracle_adapter_f64 failed ({rc})")
    grads = {"x": torch.from_numpy(gx), "conv_down.lin.weight": torch.from_numpy(gwd),
             "conv_down.bias": torch.from_numpy(gbd), "conv_up.lin.weight": torch.from_numpy(gwu),
             "conv_up.bias": torch.from_numpy(gbu)}
    if flags.learnable_scalar:
        grads["scalar"] = torch.from_numpy(gs)
    return torch.from_numpy(out), grads
