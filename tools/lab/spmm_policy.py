"""Lab builds for tools/lab/spmm_policy.sh: the library with non-temporal policy on the
SpMM's streamed-once arrays — Y stores, column ids, the segment path's gathers — one .so per combination."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
lib = os.path.join(ge.PKG, "lib")
VARIANTS = {"y": ("-DGCONV_SPMM_Y_NT=1",), "ycol": ("-DGCONV_SPMM_Y_NT=1", "-DGCONV_SPMM_COL_NT=1"),
            "ycolseg": ("-DGCONV_SPMM_Y_NT=1", "-DGCONV_SPMM_COL_NT=1", "-DGCONV_SPMM_SEG_NT=1"),
            "seg": ("-DGCONV_SPMM_SEG_NT=1",), "col": ("-DGCONV_SPMM_COL_NT=1",)}
for name, defs in VARIANTS.items():
    ge.build_library(os.path.join(lib, f"libgconv_adapter_spmm_{name}.so"), defines=defs, tag=f"_spmm_{name}")
