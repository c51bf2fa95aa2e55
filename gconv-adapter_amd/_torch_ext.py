"""Loader of the optional C++ autograd bridge (``csrc/torch_binding.cpp``).

When ``lib/_gconv_torch.so`` is present the adapter's autograd node lives in C++
(a few microseconds of host time per call); otherwise the pure-Python
``functional.GConvAdapterFunction`` is used. Both bind the SAME C ABI entry points — this
is a choice of binding, not a compute fallback."""
from __future__ import annotations

import importlib.util
import os

from . import _lib

EXT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_gconv_torch.so")
_ext = None
_tried = False


def get():
    """The extension module, or None if it is not built / disabled (GCONV_PY_BINDING=1)."""
    global _ext, _tried
    if _tried:
        return _ext
    _tried = True
    if os.environ.get("GCONV_PY_BINDING") or not os.path.exists(EXT_PATH):
        return None
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    spec = importlib.util.spec_from_file_location("_gconv_torch", EXT_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _lib.load()
    mod.load_library(_lib.LIB_PATH)
    _ext = mod
    return _ext
