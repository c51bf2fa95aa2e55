"""K0 on the host side: per-``edge_index`` graph plans and their cache.

PyG's ``GCNConv`` recomputes ``gcn_norm`` on every conv call (``cached=False``),
i.e. 2x per adapter per forward for the *same* ``edge_index`` tensor that the
model hands to all of its 5-10 adapters
(``/root/reference/src/models/inductive/gnn.py:103,115``). Here the CSR of the
propagation operator (and of its transpose for the backward pass), the
``deg^-1/2`` vector and the long-row segment lists are built once per
``edge_index`` and shared by every adapter call of the step.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Optional, Tuple

import torch

from . import _lib


class CsrView:
    """Device arrays of one CSR direction + the ``gconv_csr_t`` that points at them."""

    def __init__(self, num_nodes: int, info: _lib.GraphInfo, rowptr, col, long_rows,
                 long_seg_base, seg_long):
        self.rowptr, self.col = rowptr, col
        self.long_rows, self.long_seg_base, self.seg_long = long_rows, long_seg_base, seg_long
        self.nnz, self.n_long, self.n_seg = info.nnz, info.n_long, info.n_seg
        self.max_degree = info.max_degree
        self.struct = _lib.CsrStruct(
            num_nodes, info.nnz, rowptr.data_ptr(), col.data_ptr(), info.n_long, info.n_seg,
            long_rows.data_ptr(), long_seg_base.data_ptr(), seg_long.data_ptr())

    def ref(self):
        return C.byref(self.struct)


def _build_direction(lib, ei: torch.Tensor, n: int, normalize: bool, transpose: bool,
                     want_dis: bool, ws: torch.Tensor, stream: int
                     ) -> Tuple[CsrView, Optional[torch.Tensor]]:
    dev, e = ei.device, ei.shape[1]
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    n_long_cap = int(lib.gconv_graph_max_long_rows(e))
    n_seg_cap = int(lib.gconv_graph_max_segments(e))
    long_rows = torch.empty(n_long_cap, dtype=torch.int32, device=dev)
    long_seg_base = torch.empty(n_long_cap, dtype=torch.int32, device=dev)
    seg_long = torch.empty(n_seg_cap, dtype=torch.int32, device=dev)
    dis = torch.empty(max(n, 1), dtype=torch.float32, device=dev) if want_dis else None
    info = _lib.GraphInfo()
    st = lib.gconv_graph_prep(ei.data_ptr(), e, n, int(normalize), int(transpose),
                              rowptr.data_ptr(), col.data_ptr(), _lib.ptr(dis),
                              long_rows.data_ptr(), long_seg_base.data_ptr(),
                              seg_long.data_ptr(), ws.data_ptr(), ws.numel(),
                              C.byref(info), stream)
    _lib.check(st, "graph_prep")
    return CsrView(n, info, rowptr, col, long_rows, long_seg_base, seg_long), dis


class GraphPlan:
    """Everything the kernels need to apply S and S^T for one ``edge_index``."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, normalize: bool):
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise RuntimeError("edge_index must have shape [2, num_edges]")
        if edge_index.dtype != torch.int64:
            raise RuntimeError("edge_index must be int64 (PyG convention)")
        if not edge_index.is_cuda:
            raise RuntimeError("gconv_adapter_amd has no CPU path: edge_index must live on a HIP device")
        lib = _lib.load()
        ei = edge_index.contiguous()
        self.num_nodes, self.normalize = int(num_nodes), bool(normalize)
        self.num_edges = int(ei.shape[1])
        self.device = ei.device
        with torch.cuda.device(ei.device):
            stream = torch.cuda.current_stream().cuda_stream
            ws_bytes = int(lib.gconv_graph_prep_workspace_bytes(self.num_edges, self.num_nodes))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=ei.device)
            self.fwd, self.dis = _build_direction(lib, ei, self.num_nodes, self.normalize, False,
                                                  self.normalize, ws, stream)
            bwd, _ = _build_direction(lib, ei, self.num_nodes, self.normalize, True, False, ws,
                                      stream)
            self.symmetric = False
            if bwd.nnz == self.fwd.nnz:
                flag = torch.empty(1, dtype=torch.int32, device=ei.device)
                equal = C.c_int32(0)
                st = lib.gconv_csr_equal(self.fwd.rowptr.data_ptr(), self.fwd.col.data_ptr(),
                                         bwd.rowptr.data_ptr(), bwd.col.data_ptr(),
                                         self.num_nodes, self.fwd.nnz, flag.data_ptr(),
                                         C.byref(equal), stream)
                _lib.check(st, "csr_equal")
                self.symmetric = bool(equal.value)
            # S^T == S for symmetric edge lists (molecules emit both bond directions,
            # the transductive script calls to_undirected): reuse the forward CSR.
            self.bwd = self.fwd if self.symmetric else bwd

    @property
    def dis_ptr(self) -> Optional[int]:
        return _lib.ptr(self.dis) if self.normalize else None


class PlanCache:
    """Small LRU keyed on the identity of the ``edge_index`` storage.

    The cached entry keeps a reference to the tensor, so its memory cannot be
    recycled for a different edge list while the key is alive; an in-place edit
    bumps ``_version`` and misses."""

    def __init__(self, capacity: int = 8):
        self.capacity = capacity
        self._entries: "OrderedDict[tuple, Tuple[torch.Tensor, GraphPlan]]" = OrderedDict()
        self.hits = 0
        self.misses = 0

    @staticmethod
    def _key(edge_index: torch.Tensor, num_nodes: int, normalize: bool) -> tuple:
        return (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape),
                tuple(edge_index.stride()), str(edge_index.device), int(num_nodes), bool(normalize))

    def get(self, edge_index: torch.Tensor, num_nodes: int, normalize: bool) -> GraphPlan:
        key = self._key(edge_index, num_nodes, normalize)
        hit = self._entries.get(key)
        if hit is not None:
            self._entries.move_to_end(key)
            self.hits += 1
            return hit[1]
        self.misses += 1
        plan = GraphPlan(edge_index, num_nodes, normalize)
        self._entries[key] = (edge_index, plan)
        while len(self._entries) > self.capacity:
            self._entries.popitem(last=False)
        return plan

    def clear(self) -> None:
        self._entries.clear()


#: process-wide cache shared by every GConvAdapter instance
PLAN_CACHE = PlanCache()


def get_plan(edge_index: torch.Tensor, num_nodes: int, normalize: bool) -> GraphPlan:
    return PLAN_CACHE.get(edge_index, num_nodes, normalize)
