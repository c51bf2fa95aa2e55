"""Lab builds: the tile kernels' streamed-once output stores (out, dX) under other cache-policy bits of
raw_buffer_store (aux: 1 = sc0, 2 = nt, 16 = sc1; product: 2). python tools/lab/store_aux.py  (where hipcc runs), then
on the GPU box:  for a in 0 3 18 19; do GCONV_ADAPTER_LIB=gconv-adapter_amd/lib/libgconv_adapter_aux$a.so python tools/tilebench.py hiv hiv4096; done"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
for aux in (0, 3, 18, 19):
    ge.build_library(os.path.join(ge.PKG, "lib", f"libgconv_adapter_aux{aux}.so"), defines=(f"-DGCONV_STORE_AUX={aux}",), tag=f"_aux{aux}")
