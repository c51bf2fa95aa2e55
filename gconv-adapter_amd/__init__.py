"""gconv_adapter_amd — MI355X-native GConv-Adapter hot path.

The directory is called ``gconv-adapter_amd`` (not importable as written); the
repo-root module ``gconv_adapter_amd.py`` aliases it, so
``_target_: gconv_adapter_amd.GConvAdapter`` works in the reference's Hydra config.
"""
from .adapter import GConvAdapter, GCNConvParams
from .functional import GConvAdapterFunction, gconv_adapter, spmm, gemm_down, gemm_up_fused
from .graph import GraphPlan, PlanCache, PLAN_CACHE, get_plan, prefetch_plan
from .hydra_compat import instantiate, load_finetune_config
from . import synth, dist, optim, data, partition, gin, norm, relbias, _lib, _torch_ext
from .data import MoleculeStore
from .optim import FusedAdam
from .partition import RowSlabGConvAdapter, RowSlabs
from .gin import MolecularGINConv
from .norm import BatchNorm1d
from .relbias import add_conv_relational_bias

__all__ = [
    "GConvAdapter", "GCNConvParams", "GConvAdapterFunction", "gconv_adapter", "spmm",
    "gemm_down", "gemm_up_fused", "GraphPlan", "PlanCache", "PLAN_CACHE", "get_plan", "prefetch_plan",
    "instantiate", "load_finetune_config", "synth", "dist", "optim", "FusedAdam", "data", "MoleculeStore",
    "partition", "RowSlabGConvAdapter", "RowSlabs", "gin", "MolecularGINConv", "norm", "BatchNorm1d",
    "relbias", "add_conv_relational_bias",
]
