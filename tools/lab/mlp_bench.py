"""The frozen GIN layer's MLP: bf16x3 MFMA kernel (csrc/mlp.hip) against nn.Linear (rocBLAS fp32),
forward + input gradient, HIP events. python tools/lab/mlp_bench.py [n] [d]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13104
d = int(sys.argv[2]) if len(sys.argv) > 2 else 300
layer = ga.MolecularGINConv(d).to(dev)
for p in layer.parameters():
    p.requires_grad_(False)
a = torch.randn(n, d, device=dev, requires_grad=True)
g = torch.randn(n, d, device=dev)

def ev(fn, it=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3

def run(fn):
    def step():
        a.grad = None
        fn(a).backward(g)
    return ev(step)
flop = 2 * 2 * 2.0 * n * d * 2 * d
mine = run(lambda t: ga.gin.GinMlpFunction.apply(t, layer))
ref = run(lambda t: layer.mlp(t))
print(f"n={n} d={d}: bf16x3 {mine:.1f} us ({flop / mine / 1e6:.1f} TFLOP/s fp32-equivalent), nn.Linear {ref:.1f} us ({flop / ref / 1e6:.1f} TFLOP/s)")
