"""Device time of the adapter forward / backward through the C ABI only (no autograd, no allocator):
    python tools/tilebench.py [hiv|esol|arxiv ...]
Back-to-back launches on one stream, HIP events around 200 of them, best of 3 — the GPU-side
cost of one call, for comparing kernel variants (GCONV_ADAPTER_LIB selects the build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GCONV_PY_BINDING"] = "1"
import torch
import gconv_adapter_amd as ga
from gconv_adapter_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
if os.environ.get("GCONV_BENCH_HIDDEN"):   # lab: another hidden width for every workload (chunks per row = waves per workgroup)
    for _w in ga.synth.WORKLOADS.values():
        _w["d"] = int(os.environ["GCONV_BENCH_HIDDEN"])


def ev(fn, iters=200, warm=10, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


def run(name):
    name, _, graphs = name.partition(":")          # "hiv:2048" = 2048 molecules of the hiv generator
    batch, cfg = ga.synth.make_workload(name, graphs=int(graphs) if graphs else None)
    d, r = cfg["d"], cfg["r"]
    x = batch.x.reshape(-1, d).to(dev); ei = batch.edge_index.to(dev); g = torch.randn_like(x)
    n = x.shape[0]
    m = ga.GConvAdapter(d, r, learnable_scalar=True).to(dev)
    plan = ga.get_plan(ei, n, True)
    fp, wsf, wsb = plan.sizes(lib, d, r)
    out = torch.empty_like(x); saved = torch.empty(2, n, fp, device=dev)
    wsF = torch.zeros(max(wsf, 16), dtype=torch.uint8, device=dev)
    wsB = torch.zeros(max(wsb, 16), dtype=torch.uint8, device=dev)
    P = _lib.AdapterParams(m.conv_down.lin.weight.data_ptr(), m.conv_down.bias.data_ptr(),
                           m.conv_up.lin.weight.data_ptr(), m.conv_up.bias.data_ptr(), m.scalar.data_ptr(),
                           d, r, 1, 1)
    gx = torch.empty_like(x)
    gp = [torch.empty_like(p) for p in (m.conv_down.lin.weight, m.conv_down.bias, m.conv_up.lin.weight,
                                        m.conv_up.bias, m.scalar)]
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        rc = lib.gconv_adapter_fwd_f32(plan.ref(), C.byref(P), x.data_ptr(), out.data_ptr(), saved[0].data_ptr(),
                                       None, saved[1].data_ptr(), wsF.data_ptr(), wsf, st)
        assert rc == 0, rc

    def bwd():
        rc = lib.gconv_adapter_bwd_f32(plan.ref(), C.byref(P), x.data_ptr(), g.data_ptr(), saved[0].data_ptr(), None,
                                       saved[1].data_ptr(), gx.data_ptr(), gp[0].data_ptr(), gp[1].data_ptr(),
                                       gp[2].data_ptr(), gp[3].data_ptr(), gp[4].data_ptr(), wsB.data_ptr(), wsb, st)
        assert rc == 0, rc

    fwd(); bwd(); torch.cuda.synchronize()
    tf, tb = ev(fwd), ev(bwd)
    print(f"{name}: N={n} tiles={plan.n_tiles} d={d} r={r}  fwd {tf:7.1f} us  bwd {tb:7.1f} us  sum {tf + tb:7.1f}")


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["hiv", "esol"]):
        run(w)
