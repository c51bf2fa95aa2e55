"""gconv_adapter_amd — MI355X-native GConv-Adapter hot path.

The directory is called ``gconv-adapter_amd`` (not importable as written); the
repo-root module ``gconv_adapter_amd.py`` aliases it, so
``_target_: gconv_adapter_amd.GConvAdapter`` works in the reference's Hydra config.
"""
from .adapter import GConvAdapter, GCNConvParams
from .functional import GConvAdapterFunction, gconv_adapter, spmm, gemm_down, gemm_up_fused
from .graph import GraphPlan, PlanCache, PLAN_CACHE, get_plan, prefetch_plan, CompletionEvent
from .hydra_compat import instantiate, load_finetune_config
from . import synth, dist, optim, data, partition, gin, norm, relbias, readout, _lib, _torch_ext
from .data import MoleculeStore
from .optim import FusedAdam
from .partition import RowSlabGConvAdapter, RowSlabs
from .gin import MolecularGINConv
from .norm import BatchNorm1d, sync_batch_norm
from .relbias import add_conv_relational_bias
from .readout import global_mean_pool, global_add_pool, masked_loss



def register_ops():
    """Loads the C++ binding (if it is not loaded yet) and returns ``torch.ops.gconv``: the torch custom
    ops ``graph_plan``, ``graph_prep``, ``adapter_fwd``, ``adapter_bwd``, ``adapter`` (differentiable)
    and ``spmm`` (csrc/torch_binding.cpp, TORCH_LIBRARY(gconv, ...)). Raises if the binding is not
    built or disabled (GCONV_PY_BINDING=1)."""
    import torch
    if _torch_ext.get() is None:
        raise RuntimeError("gconv_adapter_amd: lib/_gconv_torch.so is not built (python -c 'import __graft_entry__ as g; "
                           "g.build()') or GCONV_PY_BINDING is set")
    return torch.ops.gconv


__all__ = [
    "GConvAdapter", "GCNConvParams", "GConvAdapterFunction", "gconv_adapter", "spmm",
    "gemm_down", "gemm_up_fused", "GraphPlan", "PlanCache", "PLAN_CACHE", "get_plan", "prefetch_plan", "CompletionEvent",
    "instantiate", "load_finetune_config", "synth", "dist", "optim", "FusedAdam", "data", "MoleculeStore",
    "partition", "RowSlabGConvAdapter", "RowSlabs", "gin", "MolecularGINConv", "norm", "BatchNorm1d", "sync_batch_norm",
    "relbias", "add_conv_relational_bias", "register_ops", "readout", "global_mean_pool", "global_add_pool", "masked_loss",
]
