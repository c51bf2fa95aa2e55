"""The adapter's layer_norm tail (LayerNorm + scalar) forward + backward: torch ops vs csrc/norm.hip.
    python tools/lnbench.py [N d]      (default: the 512-molecule HIV shard, 13104 x 300)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gconv_adapter_amd.functional import LayerNormScaleFunction

dev = torch.device("cuda:0")
n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (13104, 300)
z = torch.randn(n, d, device=dev, requires_grad=True)
ln = torch.nn.LayerNorm(d).to(dev)
s = torch.nn.Parameter(torch.ones(1, device=dev))
go = torch.randn(n, d, device=dev)
leaves = [z, ln.weight, ln.bias, s]


def ev(fn, iters=200, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(10):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


def ref():
    for q in leaves:
        q.grad = None
    (ln(z) * s).backward(go)


def fused():
    for q in leaves:
        q.grad = None
    LayerNormScaleFunction.apply(z, ln.weight, ln.bias, s, ln.eps).backward(go)


from gconv_adapter_amd import _torch_ext
from gconv_adapter_amd.functional import _stream
ext = _torch_ext.get()


def fused_cpp():
    for q in leaves:
        q.grad = None
    ext.layer_norm_scale(z, ln.weight, ln.bias, s, ln.eps, _stream(0)).backward(go)


print(f"[{n}, {d}] LayerNorm + scalar, fwd + bwd: torch {ev(ref):.1f} us, fused (Python binding) {ev(fused):.1f} us, "
      f"fused (C++ binding) {ev(fused_cpp):.1f} us per call")

# ---- inside a step: 5 adapters with normalization='layer_norm' on the workload, fwd + bwd
# (GCONV_TORCH_LAYER_NORM=1 python tools/lnbench.py ... runs the torch tail for comparison)
import gconv_adapter_amd as ga
ga.dist.pin_to_local_cores()
for name, norm in (("hiv", "none"), ("hiv", "layer_norm"), ("hiv4096", "none"), ("hiv4096", "layer_norm")):
    batch, cfg = ga.synth.make_workload(name)
    mods = torch.nn.ModuleList([ga.GConvAdapter(cfg["d"], cfg["r"], learnable_scalar=True, normalization=norm,
                                                normalize=True) for _ in range(5)]).to(dev)
    with torch.no_grad():
        for p in mods.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.1)
    x, ei = batch.x.to(dev), batch.edge_index.to(dev)
    gout = torch.randn_like(x)
    params = list(mods.parameters())

    def step():
        for p in params:
            p.grad = None
        h = x
        for m in mods:
            h = m(h, ei)
        h.backward(gout)

    tail = "torch LayerNorm + mul" if os.environ.get("GCONV_TORCH_LAYER_NORM") else "fused tail"
    print(f"{name}: 5 adapters, normalization={norm}, fwd + bwd, {tail}: {ev(step, iters=200):.1f} us per step")
