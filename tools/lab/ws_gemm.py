"""One f16x2 GEMM of the frozen GIN MLP through the C ABI, HIP events: the W-stationary kernel (GCONV_MLP_WS,
read once per process) or the tiled one; with --build, the leave-one-out lab libraries (GCONV_MLP_WS_LAB).
    python tools/lab/ws_gemm.py --build          (where hipcc runs)
    python tools/lab/ws_gemm.py [rows ...]       (GPU; GCONV_ADAPTER_LIB selects a lab library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LAB = {"noepi": 2, "nostore": 4, "nodma": 8, "nomfma_noepi": 3}
if "--build" in sys.argv:
    import __graft_entry__ as g
    lib = lambda name: os.path.join(ROOT, "gconv-adapter_amd", "lib", f"libgconv_adapter_ws_{name}.so")
    if "--sched" in sys.argv:     # fragment-read distance variants instead of the leave-one-out set
        for ahead in (0, 1, 3):
            g.build_library(out=lib(f"ahead{ahead}"), defines=[f"-DGCONV_MLP_WS_SCHED={ahead}"])
    else:
        for name, bits in LAB.items():
            g.build_library(out=lib(name), defines=[f"-DGCONV_MLP_WS_LAB={bits}"])
    sys.exit(0)
import torch
import gconv_adapter_amd as ga
from gconv_adapter_amd import gin
dev = torch.device("cuda:0")
rows = [int(a) for a in sys.argv[1:] if a.isdigit()] or [13104, 102952]

def ev(fn, it=30, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(5): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) / it * 1e3)
    return best

d = 300
layer = ga.MolecularGINConv(d).to(dev)
w1, b1, w2, b2 = layer.mlp[0].weight, layer.mlp[0].bias, layer.mlp[2].weight, layer.mlp[2].bias
pw = gin._PairWeights()
for n in rows:
    x = torch.randn(n, d, device=dev)
    a2 = gin._rows_pair(x)
    w1p, w2p = pw.get(w1, b1, False), pw.get(w2, b2, False)
    w2t = pw.get(w2, None, True)
    out = []
    h2 = gin._pair_gemm(a2, n, d, w1p, b1, None, 2 * d, 1, True)
    t1 = ev(lambda: gin._pair_gemm(a2, n, d, w1p, b1, None, 2 * d, 1, True))      # K 300 -> 600, ReLU, pair out
    t2 = ev(lambda: gin._pair_gemm(h2, n, 2 * d, w2p, b2, None, d, 0, False))     # K 600 -> 300, fp32 out
    t3 = ev(lambda: gin._pair_gemm(a2, n, d, w2t, None, h2, 2 * d, 2, True))      # K 300 -> 600, masked, pair out
    print(f"rows {n}: K=300 relu/pair {t1:.1f} us, K=600 fp32 {t2:.1f} us, K=300 mask/pair {t3:.1f} us", flush=True)
