"""ctypes binding of ``libgconv_adapter.so`` (the C ABI in ``include/gconv_adapter.h``).

There is no CPU compute path in this package: if the HIP library is missing or
cannot be loaded, every entry point raises ``RuntimeError`` — loudly, so a GPU
run can never pass on a silent fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# GCONV_ADAPTER_LIB: development override to load another build of the same C ABI (kernel variants)
LIB_PATH = os.environ.get("GCONV_ADAPTER_LIB") or os.path.join(_HERE, "lib", "libgconv_adapter.so")

ACT_CODES = {"none": 0, "relu": 1, "silu": 2}
SPMM_IN_PRESCALED = 1
SPMM_OUT_PRESCALE = 2
LONG_ROW = 64
SEG_EDGES = 128
TILE_NODES = 16            # GCONV_TILE_NODES: smallest nominal tile
TILE_NODES_MAX = 128       # GCONV_TILE_NODES_MAX
TILE_HALO_NODES = 64       # GCONV_TILE_HALO_NODES
TILE_HALO_EDGES = 256      # GCONV_TILE_HALO_EDGES
TILE_HALO_SLOT = 456       # GCONV_TILE_HALO_SLOT (4-byte words per tile)
NUM_CU = 256


def tile_nodes(num_nodes: int) -> int:
    """Nominal nodes per tile of a graph (``tile_nominal`` in ``csrc/graph_prep.hip`` without
    its environment overrides; ``gconv_tile_nodes`` is the library's own answer): about N / 256 —
    one tile per persistent workgroup — never below ``TILE_NODES``; beyond 256 x ``TILE_NODES_MAX``
    nodes, N / (256 k) for the smallest k that keeps it under that maximum."""
    if num_nodes <= 0:
        return TILE_NODES
    per_wg = -(-num_nodes // NUM_CU)
    k = -(-per_wg // TILE_NODES_MAX)
    return max(TILE_NODES, -(-num_nodes // (NUM_CU * k)))


def tile_slack(tile: int, num_nodes: int) -> int:
    """How far below its nominal position a cut may sit (``tile_slack`` in ``csrc/graph_prep.hip``):
    a quarter tile with one tile per workgroup, a whole tile (at most 64 nodes) with several."""
    one_per_workgroup = -(-max(num_nodes, 1) // tile) <= NUM_CU
    return min(max(tile // 4 if one_per_workgroup else tile, 1), 64)

_i32p = C.POINTER(C.c_int32)


class CsrStruct(C.Structure):
    """``gconv_csr_t``"""
    _fields_ = [
        ("num_nodes", C.c_int32),
        ("nnz", C.c_int32),
        ("rowptr", C.c_void_p),
        ("col", C.c_void_p),
        ("n_long", C.c_int32),
        ("n_seg", C.c_int32),
        ("long_rows", C.c_void_p),
        ("long_seg_base", C.c_void_p),
        ("seg_long", C.c_void_p),
        ("long_sync", C.c_void_p),
        ("near_edges", C.c_int32),
    ]


class TilingStruct(C.Structure):
    """``gconv_tiling_t``"""
    _fields_ = [("tiles", C.c_void_p), ("lcol", C.c_void_p), ("halo", C.c_void_p)]


class GraphStruct(C.Structure):
    """``gconv_graph_t``"""
    _fields_ = [
        ("fwd", CsrStruct),
        ("bwd", CsrStruct),
        ("dis", C.c_void_p),
        ("tile_ptr", C.c_void_p),
        ("tiles_fwd", TilingStruct),
        ("tiles_bwd", TilingStruct),
        ("n_tiles", C.c_int32),
        ("tile_nodes_max", C.c_int32),
        ("tile_edges_max", C.c_int32),
        ("symmetric", C.c_int32),
        ("normalize", C.c_int32),
        ("max_degree", C.c_int32),
        ("bad_index", C.c_int32),
    ]


class PendingFinalize(C.Structure):
    """``gconv_pending_finalize_t`` (opaque)"""
    _fields_ = [("opaque", C.c_uint64 * 24)]


FINALIZE_BATCH = 5
BUILD_BACKGROUND = 1   # GCONV_BUILD_BACKGROUND
ABI_VERSION = 11   # GCONV_ABI_VERSION of include/gconv_adapter.h this binding was written against


class AdapterParams(C.Structure):
    """``gconv_adapter_params_t``"""
    _fields_ = [
        ("w_down", C.c_void_p),
        ("b_down", C.c_void_p),
        ("w_up", C.c_void_p),
        ("b_up", C.c_void_p),
        ("scalar", C.c_void_p),
        ("hidden", C.c_int32),
        ("rank", C.c_int32),
        ("act", C.c_int32),
        ("skip", C.c_int32),
        ("walk", C.c_int32),
    ]


_vp, _sz, _i32, _i64, _u32 = C.c_void_p, C.c_size_t, C.c_int32, C.c_int64, C.c_uint32
_csr_p = C.POINTER(CsrStruct)
_graph_p = C.POINTER(GraphStruct)

#: name -> (restype, argtypes); must list every function declared in include/gconv_adapter.h
SIGNATURES = {
    "gconv_abi_version": (_i32, []),
    "gconv_arm_completion_event": (_i32, [C.c_void_p]),
    "gconv_abi_struct_bytes": (_i32, [_i32]),
    "gconv_status_string": (C.c_char_p, [_i32]),
    "gconv_padded_rank": (_i32, [_i32]),
    "gconv_tile_nodes": (_i32, [_i32]),
    "gconv_graph_build_bytes": (_sz, [_i64, _i32]),
    "gconv_graph_build": (_i32, [_vp, _i64, _i32, _i32, _vp, _sz, _graph_p, _vp]),
    "gconv_graph_build_ex": (_i32, [_vp, _i64, _i32, _i32, _u32, _vp, _sz, _graph_p, _vp]),
    "gconv_spmm_workspace_bytes": (_sz, [_csr_p, _i32]),
    "gconv_spmm_csr_f32": (_i32, [_csr_p, _vp, _vp, _vp, _i32, _vp, _i32, _u32, _vp, _sz, _vp]),
    "gconv_gemm_down_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "gconv_gemm_up_fused_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "gconv_adapter_fwd_workspace_bytes": (_sz, [_graph_p, _i32, _i32]),
    "gconv_adapter_bwd_workspace_bytes": (_sz, [_graph_p, _i32, _i32]),
    "gconv_adapter_fwd_f32": (_i32, [_graph_p, C.POINTER(AdapterParams), _vp, _vp, _vp, _vp,
                                     _vp, _vp, _sz, _vp]),
    "gconv_adapter_infer_workspace_bytes": (_sz, [_graph_p, _i32, _i32]),
    "gconv_adapter_infer_f32": (_i32, [_graph_p, C.POINTER(AdapterParams), _vp, _vp, _vp, _sz, _vp]),
    "gconv_adapter_bwd_f32": (_i32, [_graph_p, C.POINTER(AdapterParams), _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gconv_adapter_bwd_partial_f32": (_i32, [_graph_p, C.POINTER(AdapterParams), _vp, _vp, _vp, _vp,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                             C.POINTER(PendingFinalize), _vp]),
    "gconv_adapter_bwd_finalize_f32": (_i32, [C.POINTER(PendingFinalize), _i32, _vp]),
    "gconv_adapter_slab_fwd_workspace_bytes": (_sz, [_graph_p, _i32]),
    "gconv_adapter_slab_bwd_workspace_bytes": (_sz, [_graph_p, _i32, _i32, _i32, _i32]),
    "gconv_adapter_slab_fwd_f32": (_i32, [_graph_p, C.POINTER(AdapterParams), _i32, _i32, _i32, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gconv_adapter_slab_bwd_f32": (_i32, [_graph_p, C.POINTER(AdapterParams), _i32, _i32, _i32, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _sz, _vp]),
    "gconv_collate_workspace_bytes": (_sz, [_i32]),
    "gconv_collate_f32": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, C.c_int64, _vp, _vp, _vp, _vp,
                                 _vp, _sz, _vp]),
    "gconv_graph_from_store_bytes": (_sz, [_i32, C.c_int64, _i32]),
    "gconv_graph_from_store": (_i32, [_graph_p, _vp, _vp, _vp, _i32, _i32, C.c_int64, _i32, _vp, _sz,
                                      _graph_p, _vp]),
    "gconv_mlp_split_bytes": (_sz, [_i32, _i32]),
    "gconv_mlp_rows_split_bytes": (_sz, [_i32, _i32]),
    "gconv_mlp_split_weights_bf16x3": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "gconv_mlp_split_rows_bf16x3": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "gconv_mlp_gemm_bf16x3": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gconv_mlp_weights_pair_bytes": (_sz, [_i32, _i32]),
    "gconv_mlp_rows_pair_bytes": (_sz, [_i32, _i32]),
    "gconv_mlp_split_weights_f16x2": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "gconv_mlp_split_rows_f16x2": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "gconv_mlp_gemm_f16x2": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gconv_gin_aggregate_f16x2": (_i32, [_csr_p, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "gconv_gin_code_count_bytes": (_sz, [_i32]),
    "gconv_gin_code_count": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "gconv_gin_aggregate_f32": (_i32, [_csr_p, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "gconv_layer_norm_scale_bwd_workspace_bytes": (_sz, [_i32, _i32]),
    "gconv_layer_norm_scale_fwd_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_float, _vp]),
    "gconv_layer_norm_scale_bwd_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                                              _vp, _sz, _vp]),
    "gconv_batch_norm_workspace_bytes": (_sz, [_i32, _i32]),
    "gconv_batch_norm_fwd_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _i32, _vp, _vp, _vp,
                                        _i32, _i32, _vp, _sz, _vp]),
    "gconv_batch_norm_fwd_tracked_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _i32, _vp, _vp,
                                                _vp, _i32, _i32, _vp, _sz, _vp]),
    "gconv_batch_norm_apply_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "gconv_batch_norm_bwd_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "gconv_batch_norm_sync_words": (_sz, [_i32]),
    "gconv_batch_norm_local_stats_f32": (_i32, [_vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "gconv_batch_norm_fwd_sync_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _i32, _vp, _vp, _vp, _vp,
                                             _i32, _i32, _vp, _i32, _vp]),
    "gconv_batch_norm_bwd_local_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "gconv_batch_norm_bwd_dx_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "gconv_segment_pool_f32": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "gconv_segment_pool_bwd_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "gconv_masked_loss_f64": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "gconv_masked_loss_workspace_bytes": (_sz, [_i64]),
    "gconv_masked_loss_ws_f64": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _sz, _vp]),
    "gconv_adam_step_f32": (_i32, [_vp, _vp, _vp, _vp, C.c_int64, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_int64, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads the HIP library (once). Raises RuntimeError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"gconv_adapter_amd: HIP library not built ({LIB_PATH} is missing). Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` in the repo root; "
            "there is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise RuntimeError(f"gconv_adapter_amd: cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    mirrors = (CsrStruct, GraphStruct, AdapterParams, PendingFinalize)
    if lib.gconv_abi_version() == ABI_VERSION and any(lib.gconv_abi_struct_bytes(i) != C.sizeof(t)
                                            for i, t in enumerate(mirrors)):
        raise RuntimeError("gconv_adapter_amd: ctypes struct mirrors do not match the library's structs")
    if lib.gconv_abi_version() != ABI_VERSION:
        raise RuntimeError("gconv_adapter_amd: ABI version mismatch, rebuild the library")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    """Maps a non-zero ``gconv_status_t`` to RuntimeError (the exception type the
    reference's train/eval loops catch and skip on:
    ``scripts/finetune_inductive_learning.py:51-53,109-111``)."""
    if status != 0:
        msg = load().gconv_status_string(status).decode()
        raise RuntimeError(f"gconv_adapter_amd.{what} failed: {msg} (status {status})")


def ptr(t) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
