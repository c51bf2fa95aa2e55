"""Device-resident molecule dataset + on-device collation.

The reference collates on the CPU (PyG ``DataLoader`` -> ``Batch.from_data_list``) and copies
every batch to the device (``/root/reference/scripts/finetune_inductive_learning.py:291-293,45``).
Molecule datasets of the size the reference trains on fit in HBM many times over (HIV: 41 k
molecules, 1.3 GB of features at d = 300), so ``MoleculeStore`` keeps the whole dataset on the
device and ``collate`` builds a batch there — two launches, no per-step host <-> device traffic,
the same tensors PyG's collation produces (bit-exact: data movement and integer offsets only).
The batch's graph plan is started on the prefetch worker right away.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .functional import _stream
from .graph import GraphPlan, PLAN_CACHE, prefetch_plan
from .synth import GraphBatch


class MoleculeStore:
    """All graphs of a dataset, back to back, on one device.

    Build it from a collated batch of the WHOLE dataset (``from_batch``): node features
    ``x [N, d]``, ``edge_index [2, E]`` (global node ids, any edge order), ``graph_ptr [G + 1]``.
    """

    def __init__(self, x: torch.Tensor, node_ptr: torch.Tensor, edge_ptr: torch.Tensor,
                 edge_src: torch.Tensor, edge_dst: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("MoleculeStore keeps the dataset on a HIP device (no CPU path)")
        self.x, self.device = x.contiguous(), x.device
        self.hidden = int(x.shape[1])
        self.node_ptr_host = node_ptr.cpu().numpy().astype(np.int64)
        self.edge_ptr_host = edge_ptr.cpu().numpy().astype(np.int64)
        self.node_ptr = node_ptr.to(device=self.device, dtype=torch.int32)
        self.edge_ptr = edge_ptr.to(device=self.device, dtype=torch.int32)
        self.edge_src = edge_src.to(device=self.device, dtype=torch.int32).contiguous()
        self.edge_dst = edge_dst.to(device=self.device, dtype=torch.int32).contiguous()
        self.num_graphs = int(node_ptr.numel()) - 1
        self._lib = _lib.load()
        self._edge_index_all = None     # set by from_batch: global-id edge list of the dataset
        self._ds_plans = {}             # normalize -> (dataset GraphPlan, nnz_ptr device, nnz_ptr host)

    @classmethod
    def from_batch(cls, batch: GraphBatch, device) -> "MoleculeStore":
        """Groups the edges by graph (stable: a graph's edges keep their relative order, as
        PyG's per-graph ``Data`` objects would hold them) and makes node ids graph-local."""
        ptr = batch.graph_ptr.long()
        src, dst = batch.edge_index[0].long(), batch.edge_index[1].long()
        gid = torch.bucketize(dst, ptr[1:], right=True)
        order = torch.sort(gid, stable=True).indices
        src, dst, gid = src[order], dst[order], gid[order]
        edge_ptr = torch.zeros(ptr.numel(), dtype=torch.long)
        edge_ptr[1:] = torch.bincount(gid, minlength=ptr.numel() - 1).cumsum(0)
        store = cls(batch.x.to(device), ptr, edge_ptr, src - ptr[gid], dst - ptr[gid])
        store._edge_index_all = torch.stack([src, dst]).to(device)
        return store

    def _dataset_plan(self, normalize: bool):
        """Plan of the whole dataset's edge list (ONE gconv_graph_build, at first use): the source
        of every batch plan. None if the dataset does not qualify for the shortcut."""
        hit = self._ds_plans.get(normalize)
        if hit is None:
            hit = (None, None, None)
            if self._edge_index_all is not None and int(self.node_ptr_host[-1]) > 0:
                plan = GraphPlan(self._edge_index_all, int(self.node_ptr_host[-1]), normalize)
                if plan.symmetric and plan.fwd.n_long == 0:
                    nnz_ptr = plan.fwd.rowptr[self.node_ptr.long()].contiguous()
                    hit = (plan, nnz_ptr, nnz_ptr.cpu().numpy().astype(np.int64))
            self._ds_plans[normalize] = hit
        return hit

    def collate(self, graph_ids: Sequence[int], with_batch_vector: bool = False,
                normalize: Optional[bool] = None):
        """The batch made of ``graph_ids`` (in that order): ``GraphBatch`` on the device
        (``graph_ptr`` int32 on the device) and, optionally, the node -> graph vector. If
        ``normalize`` is given, the batch's graph plan is started on the prefetch worker."""
        ids_host = np.asarray(graph_ids, dtype=np.int64)
        if ids_host.size and (ids_host.min() < 0 or ids_host.max() >= self.num_graphs):
            raise RuntimeError("MoleculeStore.collate: graph id out of range")
        nb = int(ids_host.size)
        n_out = int((self.node_ptr_host[ids_host + 1] - self.node_ptr_host[ids_host]).sum()) if nb else 0
        e_out = int((self.edge_ptr_host[ids_host + 1] - self.edge_ptr_host[ids_host]).sum()) if nb else 0
        dev = self.device
        ids = torch.as_tensor(ids_host.astype(np.int32)).to(dev, non_blocking=True)
        x = torch.empty(n_out, self.hidden, dtype=torch.float32, device=dev)
        ei = torch.empty(2, e_out, dtype=torch.int64, device=dev)
        gptr = torch.empty(nb + 1, dtype=torch.int32, device=dev)
        bvec = torch.empty(n_out, dtype=torch.int32, device=dev) if with_batch_vector else None
        wsb = max(int(self._lib.gconv_collate_workspace_bytes(nb)), 16)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        st = self._lib.gconv_collate_f32(
            self.x.data_ptr(), self.hidden, self.node_ptr.data_ptr(), self.edge_ptr.data_ptr(),
            self.edge_src.data_ptr(), self.edge_dst.data_ptr(), ids.data_ptr(), nb, e_out, x.data_ptr(),
            ei.data_ptr(), gptr.data_ptr(), bvec.data_ptr() if bvec is not None else None,
            ws.data_ptr(), wsb, _stream(dev.index))
        _lib.check(st, "collate")
        batch = GraphBatch(x=x, edge_index=ei, graph_ptr=gptr, num_graphs=nb)
        if normalize is not None and n_out > 0:
            self._plan_for(ei, ids, ids_host, n_out, bool(normalize))
        return (batch, bvec) if with_batch_vector else batch

    def _plan_for(self, ei, ids, ids_host, n_out: int, normalize: bool) -> None:
        """Graph plan of the collated batch, registered for `ei` in the plan cache. Assembled from
        the dataset's plan (no sort, no synchronisation) when the dataset allows it, otherwise
        built from the edge list on the prefetch worker."""
        ds, nnz_ptr, nnz_host = self._dataset_plan(normalize)
        if ds is None:
            prefetch_plan(ei, n_out, normalize)
            return
        nb = int(ids_host.size)
        nnz = int((nnz_host[ids_host + 1] - nnz_host[ids_host]).sum())
        biggest = int((self.node_ptr_host[ids_host + 1] - self.node_ptr_host[ids_host]).max())
        nbytes = int(self._lib.gconv_graph_from_store_bytes(n_out, nnz, nb))
        arena = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        out = _lib.GraphStruct()
        import ctypes as C
        st = self._lib.gconv_graph_from_store(ds.ref(), self.node_ptr.data_ptr(), nnz_ptr.data_ptr(),
                                              ids.data_ptr(), nb, n_out, nnz, biggest, arena.data_ptr(), nbytes,
                                              C.byref(out), _stream(self.device.index))
        if st == -2:   # GCONV_E_UNSUPPORTED
            prefetch_plan(ei, n_out, normalize)
            return
        _lib.check(st, "graph_from_store")
        PLAN_CACHE.put(ei, n_out, normalize, GraphPlan(ei, n_out, normalize, _built=(arena, out, 0)))
