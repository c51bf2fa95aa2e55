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

import contextlib
import ctypes as C
import os
import threading
import weakref
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Tuple

import torch

from . import _lib


def _check_edge_index(edge_index: torch.Tensor) -> None:
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise RuntimeError("edge_index must have shape [2, num_edges]")
    if edge_index.dtype != torch.int64:
        raise RuntimeError("edge_index must be int64 (PyG convention)")
    if not edge_index.is_cuda:
        raise RuntimeError("gconv_adapter_amd has no CPU path: edge_index must live on a HIP device")


class CsrView:
    """Read-only view of one CSR direction inside a plan's arena (tensors are
    materialised lazily: the kernels only need the ``gconv_csr_t``)."""

    def __init__(self, plan: "GraphPlan", struct: _lib.CsrStruct):
        # weak: plan -> view -> plan would be a reference cycle, and a discarded plan (with its
        # arena) would then live until the cyclic GC runs instead of until its last use — a
        # training loop that builds a plan per batch would keep allocating fresh arenas
        self._plan, self.struct = weakref.proxy(plan), struct
        self.nnz, self.n_long, self.n_seg = struct.nnz, struct.n_long, struct.n_seg
        #: stored edges whose column lies in its row's eighth of the node range (radix builder; 0 = not counted)
        self.near_edges = struct.near_edges
        self.addr = C.addressof(struct)   # (the gconv_csr_t inside the plan's struct: for the C++ binding)

    def ref(self):
        return C.byref(self.struct)

    def _view(self, ptr: int, count: int, dtype=torch.int32) -> torch.Tensor:
        return self._plan.arena_view(ptr, count, dtype)

    @property
    def rowptr(self) -> torch.Tensor:
        return self._view(self.struct.rowptr, self._plan.num_nodes + 1)

    @property
    def col(self) -> torch.Tensor:
        return self._view(self.struct.col, self.nnz)

    @property
    def long_rows(self) -> torch.Tensor:
        return self._view(self.struct.long_rows, self.n_long)

    @property
    def max_degree(self) -> int:
        return self._plan.struct.max_degree


_ARENA_BYTES = {}                      # (num_edges, num_nodes) -> gconv_graph_build_bytes
_NO_GUARD = contextlib.nullcontext()


class GraphPlan:
    """Everything the kernels need to apply S and S^T for one ``edge_index``:
    ONE ``gconv_graph_build`` call, ONE device allocation (the arena)."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, normalize: bool, _built=None,
                 background: bool = False):
        """``background``: the build runs on a side stream next to other kernels (GCONV_BUILD_BACKGROUND:
        no grid-wide launch — see include/gconv_adapter.h)."""
        _check_edge_index(edge_index)
        lib = _lib.load()
        ei = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
        self.num_nodes, self.normalize = int(num_nodes), bool(normalize)
        self.num_edges = int(ei.shape[1])
        self.device = ei.device
        self.struct = _lib.GraphStruct()
        if _built is not None:   # (arena, gconv_graph_t bytes | GraphStruct, status): built elsewhere
            self.arena, raw, st = _built
            if isinstance(raw, _lib.GraphStruct):
                C.memmove(C.addressof(self.struct), C.addressof(raw), C.sizeof(self.struct))
            else:
                C.memmove(C.addressof(self.struct), raw, C.sizeof(self.struct))
        else:
            key = (self.num_edges, self.num_nodes)
            nbytes = _ARENA_BYTES.get(key)
            if nbytes is None:
                if len(_ARENA_BYTES) > 4096:
                    _ARENA_BYTES.clear()
                nbytes = _ARENA_BYTES[key] = int(lib.gconv_graph_build_bytes(self.num_edges, self.num_nodes))
            # (the device guard costs ~3 us of a 60 us build: only when the tensor lives elsewhere)
            guard = torch.cuda.device(ei.device) if ei.device.index != torch.cuda.current_device() else _NO_GUARD
            with guard:
                self.arena = torch.empty(nbytes, dtype=torch.uint8, device=ei.device)
                st = lib.gconv_graph_build_ex(ei.data_ptr(), self.num_edges, self.num_nodes,
                                              int(self.normalize), _lib.BUILD_BACKGROUND if background else 0,
                                              self.arena.data_ptr(), nbytes, C.byref(self.struct),
                                              torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "graph_build")
        self.symmetric = bool(self.struct.symmetric)
        self.fwd = CsrView(self, self.struct.fwd)
        # S^T == S for symmetric edge lists (molecules emit both bond directions, the
        # transductive script calls to_undirected): the backward reuses the forward CSR.
        self.bwd = self.fwd if self.symmetric else CsrView(self, self.struct.bwd)
        self.n_tiles, self.tile_nodes_max = self.struct.n_tiles, self.struct.tile_nodes_max
        self._ref = C.byref(self.struct)
        self.struct_addr = C.addressof(self.struct)
        self._sizes = {}
        self._walk = 0

    def sizes(self, lib, d: int, r: int):
        """(padded rank, forward workspace bytes, backward workspace bytes) for hidden d and
        rank r — cached: every adapter of the step asks for the same pair."""
        hit = self._sizes.get((d, r))
        if hit is None:
            fp = int(lib.gconv_padded_rank(r))
            if fp == 0:
                raise RuntimeError(f"bottleneck_size={r} is unsupported (1..64)")
            hit = (fp, max(int(lib.gconv_adapter_fwd_workspace_bytes(self._ref, d, r)), 16),
                   max(int(lib.gconv_adapter_bwd_workspace_bytes(self._ref, d, r)), 16))
            self._sizes[(d, r)] = hit
        return hit

    def next_walk(self) -> int:
        """``gconv_adapter_params_t::walk`` for the next adapter call on this plan: 0, 1, 0, 1, ... in call order, so
        that consecutive adapters of a model forward alternate (a training batch has a fresh plan: adapter k of the
        model gets k & 1 every step). ``GConvAdapter.walk = 0 | 1`` pins it per module instead."""
        w = self._walk
        self._walk = w ^ 1
        return w

    def infer_bytes(self, lib, d: int, r: int) -> int:
        """Workspace bytes of the forward-only call (``gconv_adapter_infer_f32``) — cached like ``sizes``."""
        hit = self._sizes.get(("infer", d, r))
        if hit is None:
            hit = max(int(lib.gconv_adapter_infer_workspace_bytes(self._ref, d, r)), 16)
            self._sizes[("infer", d, r)] = hit
        return hit

    def ref(self):
        return self._ref

    def arena_view(self, ptr: int, count: int, dtype) -> torch.Tensor:
        off = ptr - self.arena.data_ptr()
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return self.arena[off:off + nbytes].view(dtype)

    @property
    def dis(self) -> Optional[torch.Tensor]:
        if not self.normalize:
            return None
        return self.arena_view(self.struct.dis, max(self.num_nodes, 1), torch.float32)

    @property
    def tile_ptr(self) -> torch.Tensor:
        return self.arena_view(self.struct.tile_ptr, self.n_tiles + 1, torch.int32)

    #: columns of a tile record (``gconv_tile_t``)
    TILE_FIELDS = ("first", "count", "n1", "n0", "slot", "e0", "ne_own", "ne_halo")

    def tiles(self, transposed: bool = False) -> torch.Tensor:
        """``[n_tiles, 8]`` tile records (``gconv_tile_t``: first, count, n1, n0, slot, e0, ne_own,
        ne_halo) of S (or of S^T), largest tile (rows incl. halo) first."""
        t = self.struct.tiles_bwd if transposed else self.struct.tiles_fwd
        return self.arena_view(t.tiles, 8 * self.n_tiles, torch.int32).view(-1, 8)

    def tile_lcol(self, transposed: bool = False) -> torch.Tensor:
        """The CSR's ``col`` in tile-local row numbers (``gconv_tiling_t.lcol``)."""
        t, csr = (self.struct.tiles_bwd, self.bwd) if transposed else (self.struct.tiles_fwd, self.fwd)
        return self.arena_view(t.lcol, csr.nnz, torch.int32)

    def tile_halo(self, transposed: bool = False) -> torch.Tensor:
        """``[n_tiles, TILE_HALO_SLOT]`` halo slots in tile order (``gconv_tiling_t.halo``; only a
        slot's used prefixes are defined)."""
        t = self.struct.tiles_bwd if transposed else self.struct.tiles_fwd
        return self.arena_view(t.halo, _lib.TILE_HALO_SLOT * self.n_tiles, torch.int32).view(-1, _lib.TILE_HALO_SLOT)

    @property
    def tile_order(self) -> torch.Tensor:
        """The size-sorted tile records of S (see ``tiles``)."""
        return self.tiles(False)

    @property
    def dis_ptr(self) -> Optional[int]:
        return self.struct.dis if self.normalize else None


class PlanCache:
    """Small LRU keyed on the identity of the ``edge_index`` storage.

    The cached entry keeps a reference to the tensor, so its memory cannot be
    recycled for a different edge list while the key is alive; an in-place edit
    bumps ``_version`` and misses.

    ``prefetch`` builds a plan for a FUTURE batch on a side stream from a worker thread, the
    way the reference's DataLoader workers collate the next batch while the current step
    runs: the graph build then overlaps the adapters of the current step instead of preceding
    those of the next one. With the C++ extension the worker is a native thread
    (``csrc/torch_binding.cpp``, no interpreter lock involved); without it, a Python thread
    (ctypes releases the GIL inside the C ABI)."""

    def __init__(self, capacity: int = 8):
        self.capacity = capacity
        self._entries: "OrderedDict[tuple, Tuple[torch.Tensor, GraphPlan]]" = OrderedDict()
        self._pending = {}
        self._lock = threading.Lock()
        self._pool: Optional[ThreadPoolExecutor] = None
        self._side_streams = {}
        self.hits = 0
        self.misses = 0

    @staticmethod
    def _key(edge_index: torch.Tensor, num_nodes: int, normalize: bool) -> tuple:
        # data_ptr is unique among live allocations of all devices; the entry keeps the
        # tensor alive, so the pointer cannot be recycled while the key exists
        return (edge_index.data_ptr(), edge_index._version, edge_index.shape[1],
                edge_index.stride(0), num_nodes, normalize)

    def _store(self, key, edge_index, plan) -> None:
        self._entries[key] = (edge_index, plan)
        while len(self._entries) > self.capacity:
            self._entries.popitem(last=False)

    def get(self, edge_index: torch.Tensor, num_nodes: int, normalize: bool) -> GraphPlan:
        key = self._key(edge_index, num_nodes, normalize)
        hit = self._entries.get(key)
        if hit is not None:
            self._entries.move_to_end(key)
            self.hits += 1
            return hit[1]
        fut = self._pending.pop(key, None)
        if isinstance(fut, tuple):                # (job id, edge_index) of the native worker
            from . import _torch_ext
            # plan_take returns once the worker has seen the side stream drain (the tile size sort is queued
            # behind the build's host synchronisation; the worker waits for it too): nothing to order here
            plan = GraphPlan(fut[1], num_nodes, normalize, _built=_torch_ext.get().plan_take(
                fut[0], torch._C._cuda_getCurrentRawStream(fut[1].device.index)))
            self.hits += 1
        elif fut is not None:                     # built (or being built) by the Python prefetcher
            plan = fut.result()
            # the arena was allocated on the side stream but is consumed on this one, and the
            # build's last launch (tile size sort) may still be queued there
            cur = torch.cuda.current_stream(plan.device)
            cur.wait_stream(self._side_streams[plan.device])
            plan.arena.record_stream(cur)
            self.hits += 1
        else:
            self.misses += 1
            plan = GraphPlan(edge_index, num_nodes, normalize)
        self._store(key, edge_index, plan)
        return plan

    def put(self, edge_index: torch.Tensor, num_nodes: int, normalize: bool, plan: GraphPlan) -> None:
        """Registers a plan that was built elsewhere (``data.MoleculeStore``) for this tensor."""
        self._store(self._key(edge_index, num_nodes, normalize), edge_index, plan)

    def prefetch(self, edge_index: torch.Tensor, num_nodes: int, normalize: bool,
                 ready: Optional["CompletionEvent"] = None) -> None:
        """Starts building the plan of a future batch; returns immediately. The build's side stream waits for what
        the current stream holds now (an event recorded here) — or, with ``ready``, for that event alone: a
        ``CompletionEvent`` armed in front of an adapter forward that has already been called (nothing is queued on
        the current stream then)."""
        key = self._key(edge_index, num_nodes, normalize)
        if key in self._entries or key in self._pending:
            return
        _check_edge_index(edge_index)
        dev = edge_index.device
        from . import _torch_ext
        ext = _torch_ext.get()
        with self._lock:
            if self._pool is None and ext is None:
                self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="gconv-plan")
            side = self._side_streams.get(dev)
            if side is None:
                side = self._side_streams[dev] = torch.cuda.Stream(device=dev)
        if ext is not None:
            ei = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
            with torch.cuda.device(dev):
                job = ext.plan_submit(ei, int(num_nodes), bool(normalize),
                                      torch._C._cuda_getCurrentRawStream(dev.index), side.cuda_stream,
                                      ready.handle if ready is not None and ready.used else 0)
            self._pending[key] = (job, ei, ready)   # (the event outlives the worker's wait for it)
            return
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))   # edge_index is complete after this point

        def build():
            with torch.cuda.device(dev), torch.cuda.stream(side):
                side.wait_event(ready)
                return GraphPlan(edge_index, num_nodes, normalize, background=True)

        self._pending[key] = self._pool.submit(build)

    def discard(self, edge_index: torch.Tensor, num_nodes: int, normalize: bool) -> None:
        """Drops the plan of a batch that will not be seen again."""
        self._entries.pop(self._key(edge_index, num_nodes, normalize), None)

    def clear(self) -> None:
        for fut in list(self._pending.values()):
            if isinstance(fut, tuple):
                from . import _torch_ext
                _torch_ext.get().plan_take(fut[0], torch._C._cuda_getCurrentRawStream(fut[1].device.index))
            else:
                fut.result()
        self._pending.clear()
        self._entries.clear()


#: process-wide cache shared by every GConvAdapter instance
PLAN_CACHE = PlanCache()


def get_plan(edge_index: torch.Tensor, num_nodes: int, normalize: bool) -> GraphPlan:
    return PLAN_CACHE.get(edge_index, num_nodes, normalize)


#: (edge_index key, copies) -> (edge_index, its block-diagonal repetition); the repetition must keep ONE identity
#: per source list, or every forward of a [B, N, d] input would miss the plan cache
_REPLICAS: "OrderedDict[tuple, Tuple[torch.Tensor, torch.Tensor]]" = OrderedDict()


def replicated_edge_index(edge_index: torch.Tensor, num_nodes: int, copies: int) -> torch.Tensor:
    """``edge_index`` of ``copies`` disjoint copies of the graph (copy b shifted by ``b * num_nodes``), cached per
    source list. ``[B, N, d]`` inputs with B > 1 (PyG's ``node_dim = -2``: every batch entry is propagated over the
    same graph, /root/reference/src/finetune/gconv_adapter.py:82-92) run as ONE call over the block-diagonal graph of
    ``B * N`` nodes instead of B calls: a copy's rows only reference that copy's rows, so the result per entry is the
    2-D result."""
    key = PlanCache._key(edge_index, num_nodes, True)[:-1] + (copies,)
    hit = _REPLICAS.get(key)
    if hit is not None:
        _REPLICAS.move_to_end(key)
        return hit[1]
    _check_edge_index(edge_index)
    if edge_index.numel():
        # an endpoint >= N would land in the NEXT copy instead of failing the graph build: checked here, once per
        # cached list (one host synchronisation)
        lo, hi = torch.aminmax(edge_index)
        if int(lo) < 0 or int(hi) >= num_nodes:
            raise RuntimeError("gconv_graph_build: edge_index entry outside [0, num_nodes) (GCONV_E_BAD_INDEX)")
    shift = torch.arange(copies, device=edge_index.device, dtype=edge_index.dtype) * num_nodes
    rep = (edge_index.unsqueeze(1) + shift.view(1, copies, 1)).reshape(2, -1)   # [2, B, E] -> [2, B * E], copy-major
    _REPLICAS[key] = (edge_index, rep)
    while len(_REPLICAS) > 4:
        _REPLICAS.popitem(last=False)
    return rep


def prefetch_plan(edge_index: torch.Tensor, num_nodes: int, normalize: bool = True,
                  ready: Optional["CompletionEvent"] = None) -> None:
    """Overlap the graph build of the NEXT batch with the current step (see PlanCache)."""
    PLAN_CACHE.prefetch(edge_index, num_nodes, normalize, ready)


class CompletionEvent:
    """A HIP event that the NEXT adapter forward of this thread records at the completion of its last kernel
    (``gconv_arm_completion_event``, ``include/gconv_adapter.h``): on the fused tile path it rides on the kernel's own
    completion signal, so nothing is queued between the forward pass and the backward pass — an event recorded
    behind the forward is a barrier packet in front of the first backward kernel (6.5 us of idle GPU per step of a
    whole 4096-molecule batch, ``tools/lab/r05_timeline.sh``). Use::

        ev = CompletionEvent()
        ...
        ev.arm()                      # in front of the forward pass's LAST adapter call
        h = adapter(h, edge_index)
        prefetch_plan(next_edge_index, n, normalize, ready=ev)   # the build runs beside the backward pass

    Needs the native binding (``_torch_ext``); without it ``arm`` does nothing and ``prefetch_plan`` records its own
    event as before. One event serves every step (re-armed, re-recorded)."""

    def __init__(self):
        from . import _torch_ext
        self._ext = _torch_ext.get()
        self.handle = self._ext.event_create() if self._ext is not None else 0
        self.used = False

    def arm(self) -> None:
        if self._ext is not None:
            self._ext.arm_completion_event(self.handle)
            self.used = True

    def done(self) -> bool:
        return self._ext is None or not self.used or bool(self._ext.event_query(self.handle))

    def __del__(self):
        if getattr(self, "_ext", None) is not None and self.handle:
            try:
                self._ext.event_destroy(self.handle)
            except Exception:
                pass
            self.handle = 0
