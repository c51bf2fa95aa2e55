"""Per-phase timing of the fused tile kernels (diagnostic build with -DGCONV_TILE_STAMPS).
    python tools/tilestamps.py [hiv|esol]
Builds lib/libgconv_adapter_stamps.so on first use (hipcc), runs one adapter forward+backward through the
ctypes binding and prints, for the largest and some median tiles, the time spent in each phase."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GCONV_PY_BINDING"] = "1"
import numpy as np, torch
import gconv_adapter_amd as ga
from gconv_adapter_amd import _lib
alt = os.environ.get("GCONV_STAMPS_LIB") or _lib.LIB_PATH.replace("libgconv_adapter.so", "libgconv_adapter_stamps.so")
if "--build-only" in sys.argv or not os.path.exists(alt):   # (build where hipcc cross-compiles: --build-only; incremental)
    import __graft_entry__ as ge
    ge.build_library(alt, defines=("-DGCONV_TILE_STAMPS",), tag="_stamps")
if "--build-only" in sys.argv:
    sys.exit(0)
_lib.LIB_PATH = alt
if os.environ.get("GCONV_BENCH_HIDDEN"):   # lab: another hidden width for every workload (chunks per row = waves per workgroup)
    for _w in ga.synth.WORKLOADS.values():
        _w["d"] = int(os.environ["GCONV_BENCH_HIDDEN"])
dev = torch.device("cuda:0")
batch, cfg = ga.synth.make_workload(next((a for a in sys.argv[1:] if not a.startswith("-")), "hiv"))
d, r = cfg["d"], cfg["r"]
x = batch.x.to(dev); ei = batch.edge_index.to(dev); g = torch.randn_like(x)
m = ga.GConvAdapter(d, r, learnable_scalar=True).to(dev)
plan = ga.get_plan(ei, x.shape[0], True)
lib = _lib.load()
fp, wsf, wsb = plan.sizes(lib, d, r)
out = torch.empty_like(x); saved = torch.empty(2, x.shape[0], fp, device=dev)
wsF = torch.zeros(wsf, dtype=torch.uint8, device=dev); wsB = torch.zeros(wsb, dtype=torch.uint8, device=dev)
P = _lib.AdapterParams(m.conv_down.lin.weight.data_ptr(), m.conv_down.bias.data_ptr(), m.conv_up.lin.weight.data_ptr(),
                       m.conv_up.bias.data_ptr(), m.scalar.data_ptr(), d, r, 1, 1)
gx = torch.empty_like(x); gp = [torch.empty_like(p) for p in (m.conv_down.lin.weight, m.conv_down.bias, m.conv_up.lin.weight, m.conv_up.bias, m.scalar)]
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    lib.gconv_adapter_fwd_f32(plan.ref(), C.byref(P), x.data_ptr(), out.data_ptr(), saved[0].data_ptr(), None, saved[1].data_ptr(), wsF.data_ptr(), wsf, st)
    lib.gconv_adapter_bwd_f32(plan.ref(), C.byref(P), x.data_ptr(), g.data_ptr(), saved[0].data_ptr(), None, saved[1].data_ptr(), gx.data_ptr(),
                              gp[0].data_ptr(), gp[1].data_ptr(), gp[2].data_ptr(), gp[3].data_ptr(), gp[4].data_ptr(), wsB.data_ptr(), wsb, st)
torch.cuda.synchronize()
grid = min(plan.n_tiles, int(os.environ.get("GCONV_STAMPS_GRID", "256")))   # (512 for a -DGCONV_TILE_WGPC=2 build)
ENTRY = {}
def show(tag, ws, off):
    st = ws[off:off + grid * 128].view(torch.int64).cpu().numpy().reshape(-1, 16)
    t0 = st[:, 0].min(); nt = st[:, 7]
    last = 6 if tag == "bwd" else 5
    print(f"{tag}: tiles {plan.n_tiles} grid {grid} max nodes {nt.max()} span {(st[:, last].max() - t0) / 100.0:.1f} us; starts: "
          f"{np.percentile((st[:,0]-t0)/100.0,[0,50,90,100]).round(1)}")
    print(f"     kernel entry -> first tile: {np.percentile((st[:,0]-st[:,8])/100.0,[0,50,100]).round(1)} us; first entry at {(st[:,8].min()-t0)/100.0:.1f}, "
          f"last end at {(st[:, last].max()-t0)/100.0:.1f} (relative to the first workgroup's stamp 0)")
    ENTRY[tag] = (st[:, 8].min(), st[:, last].max())
    # where the workgroups ran (slot 12 = HW_ID, slot 13 = XCC_ID): workgroups per CU, and how many of a CU's
    # workgroups were resident at the same time (their [entry, last stamp] intervals overlap)
    hw, xcc = st[:, 12], st[:, 13] & 0xF
    cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    per_cu, overlap = {}, 0
    for i in range(len(cu)):
        per_cu.setdefault(int(cu[i]), []).append((st[i, 8], st[i, last]))
    for ivs in per_cu.values():
        overlap += sum(1 for a in range(len(ivs)) for b in range(a + 1, len(ivs))
                       if ivs[a][0] < ivs[b][1] and ivs[b][0] < ivs[a][1])
    sizes = np.bincount([len(v) for v in per_cu.values()])
    print(f"     placement: {len(per_cu)} distinct CUs for {len(cu)} workgroups; CUs by workgroup count {dict(enumerate(sizes.tolist()))}; "
          f"co-resident pairs {overlap}")
    if tag == "fwd": print("     prologue pieces (median us): entry->9 %.2f, stage_weights %.2f, fragment loads %.2f, barrier->0 %.2f" % tuple(np.median((st[:, b] - st[:, a]) / 100.0) for a, b in ((8, 9), (9, 10), (10, 11), (11, 0))))
    print("     nt | start | setup  ph1   ph2   ph3   ph4  " + ("tail" if tag == "bwd" else ""))
    order = np.argsort(-nt)
    for i in list(order[:4]) + list(order[len(order) // 2: len(order) // 2 + 3]):
        dur = (st[i, 1:last + 1] - st[i, 0:last]) / 100.0
        print("  %5d | %5.1f | " % (nt[i], (st[i, 0] - t0) / 100.0) + " ".join("%5.1f" % v for v in dur))
show("fwd", wsF, 0)
tw = wsb - (1 << 20)
show("bwd", wsB, tw)

print("kernel boundary: forward's last stamp -> backward's first entry: %.1f us" % ((ENTRY["bwd"][0] - ENTRY["fwd"][1]) / 100.0))
