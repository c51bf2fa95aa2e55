"""ms/step of the headline workload per chunk of 200 steps from process start (a fresh box, the driver's situation):
    [GCONV_BENCH_PREFETCH_DEPTH=1|2|3] python tools/lab/settle_trace.py [seconds=12] [workload=hiv4096]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
name = sys.argv[2] if len(sys.argv) > 2 else "hiv4096"
shard = None
if os.environ.get("GCONV_LAB_SHARD"):   # "0/2": rank 0's share of the batch cut over 2 ranks (a point of the strong-scaling curve)
    r, w = os.environ["GCONV_LAB_SHARD"].split("/")
    shard = (int(r), int(w))
# STATIC: one batch, its plan cached; STATIC + ROTATE: four batches in turn, their four plans cached (no build, the step's data rotate)
stream = bench.BatchStream(name, 0, dev, k=4, fresh=not os.environ.get("GCONV_LAB_STATIC"), shard=shard,
                           rotate=bool(os.environ.get("GCONV_LAB_ROTATE")))
mods = list(bench.build_adapters(stream.cfg, True, dev))
params = [p for m in mods for p in m.parameters()]
bench.pin_host_threads()
t0 = time.perf_counter()
out = []
while time.perf_counter() - t0 < secs:
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(200):
        bench.adapter_step(mods, params, stream, True, None)
    torch.cuda.synchronize()
    out.append(((time.perf_counter() - a) / 200 * 1e3, time.perf_counter() - t0))
if os.environ.get("GCONV_LAB_SUMMARY"):
    v = sorted(ms for ms, _ in out[1:])
    print(f"{name} depth={stream.depth} bg={os.environ.get('GCONV_BG_BUILD', 'default')}: min {v[0]:.4f} median {v[len(v) // 2]:.4f} "
          f"p90 {v[int(len(v) * 0.9)]:.4f} max {v[-1]:.4f} ms/step over {len(v)} chunks of 200")
else:
    print(" ".join(f"{ms:.3f}@{t:.1f}s" for ms, t in out))
