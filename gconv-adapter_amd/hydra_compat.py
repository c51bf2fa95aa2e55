"""Hydra-style construction of the adapter.

The reference builds adapters with ``hydra.utils.instantiate(cfg.finetune.adapter,
hidden_size=emb_dim)`` (``/root/reference/src/models/inductive/gnn.py:79,82``,
``src/models/transductive/nodeformer.py:371,374``, ``difformer.py:197,200``) from
``configs/finetune/gconv_adapter.yaml``. hydra / omegaconf are used when they are
installed; otherwise this ``_target_`` resolver covers the subset the reference
needs (dotted path + kwargs + call-site overrides).
"""
from __future__ import annotations

import importlib
import os
from typing import Any, Mapping

import yaml

_CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def _locate(path: str):
    module, _, attr = path.rpartition(".")
    if not module:
        raise ValueError(f"_target_ '{path}' is not a dotted path")
    return getattr(importlib.import_module(module), attr)


def instantiate(config: Mapping[str, Any], *args, **overrides):
    """``hydra.utils.instantiate`` for flat ``_target_`` configs."""
    try:  # pragma: no cover - hydra is absent in the build image
        from hydra.utils import instantiate as _hydra_instantiate
        return _hydra_instantiate(config, *args, **overrides)
    except ImportError:
        pass
    cfg = dict(config)
    target = cfg.pop("_target_", None)
    if target is None:
        raise ValueError("config has no _target_")
    cfg.update(overrides)
    fn = _locate(target) if isinstance(target, str) else target
    return fn(*args, **cfg)


def load_finetune_config(name: str = "gconv_adapter") -> dict:
    """Loads ``configs/finetune/<name>.yaml`` (same schema as the reference's
    ``configs/finetune/gconv_adapter.yaml:1-12``: adapter / positions / type)."""
    with open(os.path.join(_CONFIG_DIR, "finetune", f"{name}.yaml")) as fh:
        return yaml.safe_load(fh)
