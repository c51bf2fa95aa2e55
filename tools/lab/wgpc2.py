"""Lab builds for tools/lab/wgpc2.sh: the library with -DGCONV_TILE_WGPC=2, plain and with phase stamps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
lib = os.path.join(ge.PKG, "lib")
ge.build_library(os.path.join(lib, "libgconv_adapter_wgpc2.so"), defines=("-DGCONV_TILE_WGPC=2",), tag="_wgpc2")
ge.build_library(os.path.join(lib, "libgconv_adapter_wgpc2_stamps.so"),
                 defines=("-DGCONV_TILE_WGPC=2", "-DGCONV_TILE_STAMPS"), tag="_wgpc2_stamps")
