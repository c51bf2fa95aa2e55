import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gconv_adapter_amd as ga
from gconv_adapter_amd import _lib
from gconv_adapter_amd.functional import GConvAdapterFunction, _stream, _ptr
dev = torch.device("cuda:0")
batch, cfg = ga.synth.make_workload("esol")
d, r = cfg["d"], cfg["r"]
x = batch.x.to(dev); ei = batch.edge_index.to(dev)
m = ga.GConvAdapter(d, r, learnable_scalar=True).to(dev)
plan = ga.get_plan(ei, x.shape[0], True)
lib = _lib.load()

def t(fn, n=2000):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6; torch.cuda.synchronize(); return dt

class Triv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w): ctx.save_for_backward(x, w); return x
    @staticmethod
    def backward(ctx, g): return g, None

w = m.conv_down.lin.weight
xg = x.clone().requires_grad_(True)
print("torch.empty_like(x)            %.1f us" % t(lambda: torch.empty_like(x)))
print("torch.empty(3,n,4)             %.1f us" % t(lambda: torch.empty(3, x.shape[0], 4, dtype=torch.float32, device=dev)))
print("trivial Function.apply         %.1f us" % t(lambda: Triv.apply(xg, w)))
print("_stream()                      %.2f us" % t(lambda: _stream(0)))
print("plan.sizes                     %.2f us" % t(lambda: plan.sizes(lib, d, r)))
print("get_plan (hit)                 %.2f us" % t(lambda: ga.get_plan(ei, x.shape[0], True)))
p = m._flat_params if hasattr(m, "_flat_params") else None
out = torch.empty_like(x); fp, wsf, wsb = plan.sizes(lib, d, r)
saved = torch.empty(2, x.shape[0], fp, device=dev); ws = torch.empty(wsf, dtype=torch.uint8, device=dev)
wd, bd, wu, bu, s = m.conv_down.lin.weight, m.conv_down.bias, m.conv_up.lin.weight, m.conv_up.bias, m.scalar
def mk(): return _lib.AdapterParams(wd.data_ptr(), _ptr(bd), wu.data_ptr(), _ptr(bu), _ptr(s), d, r, 1, 1)
print("AdapterParams(...)             %.2f us" % t(mk))
params = mk()
def call():
    return lib.gconv_adapter_fwd_f32(plan.ref(), C.byref(params), x.data_ptr(), out.data_ptr(), saved[0].data_ptr(), None,
                                     saved[1].data_ptr(), ws.data_ptr(), wsf, _stream(0))
print("ctypes gconv_adapter_fwd_f32   %.1f us (incl. 1 launch)" % t(call))
print("module forward (no grad)       %.1f us" % t(lambda: m(x, ei)))
print("module forward (x requires)    %.1f us" % t(lambda: m(xg, ei)))
with torch.no_grad():
    print("module forward (torch.no_grad) %.1f us" % t(lambda: m(x, ei)))
