"""The frozen GIN layer's MLP: the f16x2 (csrc/mlp_f16.hip) and bf16x3 (csrc/mlp.hip) MFMA kernels against
nn.Linear (rocBLAS fp32), forward + input gradient, HIP events; and the whole frozen layer (aggregation + MLP)
as one node against the round-3 composition. python tools/lab/mlp_bench.py [n] [d]"""
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

def ev(fn, it=30, reps=5):   # best of `reps` averages (box-to-box and run-to-run spread is ~15 %)
    best = 1e9
    for _ in range(reps):
        for _ in range(5): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) / it * 1e3)
    return best

def run(fn):
    def step():
        a.grad = None
        fn(a).backward(g)
    return ev(step)
flop = 2 * 2 * 2.0 * n * d * 2 * d
import time
t_end = time.perf_counter() + 1.0   # a fresh process runs its first second at ramp-up speed (bench.py: settle)
while time.perf_counter() < t_end:
    a.grad = None
    ga.gin.GinPairMlpFunction.apply(a, layer).backward(g)
torch.cuda.synchronize()
pair = run(lambda t: ga.gin.GinPairMlpFunction.apply(t, layer))
mine = run(lambda t: ga.gin.GinMlpFunction.apply(t, layer))
ref = run(lambda t: layer.mlp(t))
print(f"n={n} d={d}: f16x2 {pair:.1f} us ({flop / pair / 1e6:.1f} TFLOP/s fp32-equivalent), bf16x3 {mine:.1f} us "
      f"({flop / mine / 1e6:.1f}), nn.Linear {ref:.1f} us ({flop / ref / 1e6:.1f} TFLOP/s)")
if d == 300:   # the whole frozen layer on a molecule batch of about n nodes
    batch, cfg = ga.synth.make_workload("hiv", graphs=max(1, round(n / 25.6)))
    ei = batch.edge_index.to(dev)
    e = ei.shape[1]
    gen = torch.Generator().manual_seed(0)
    ea = torch.stack([torch.randint(0, 4, (e,), generator=gen), torch.randint(0, 3, (e,), generator=gen)], 1).to(dev)
    x = batch.x.to(dev).requires_grad_(True)
    gg = torch.randn_like(x)
    def layer_step():
        x.grad = None
        layer(x, ei, ea).backward(gg)
    one = ev(layer_step)
    ga.gin.MLP_BF16X3 = True
    old = ev(layer_step)
    ga.gin.MLP_BF16X3 = False
    print(f"frozen layer on {batch.num_nodes} nodes, fwd + bwd: one node (aggregate -> pair, f16x2) {one:.1f} us; "
          f"round 3 (aggregate fp32, split, bf16x3) {old:.1f} us")
