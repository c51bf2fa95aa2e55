"""Whole-batch step (four batches in turn, plans cached: no graph build) with a 300 us side-stream neighbour of the
background builder's shape beside the backward pass: none / ALU only / cached random 64-byte reads / device-scope ones /
random atomics (tools/lab/neighbour_probe.hip). ms/step medians per neighbour."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "neighbour_probe.so"))
lib.probe_launch.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
stream = bench.BatchStream("hiv4096", 0, dev, k=4, fresh=False, rotate=True)
mods = list(bench.build_adapters(stream.cfg, True, dev))
params = [p for m in mods for p in m.parameters()]
bench.pin_host_threads()
side = torch.cuda.Stream(device=dev)
table = torch.zeros(8 << 20, dtype=torch.uint8, device=dev)
out = torch.zeros(16, dtype=torch.int32, device=dev)
usec = float(os.environ.get("PROBE_USEC", "300"))
blocks = int(os.environ.get("PROBE_BLOCKS", "64"))

def step(kind):
    batch, x, ei, gout = stream.next(True)
    ga.optim.zero_grad(params)
    h = x
    for m in mods:
        h = m(h, ei)
    if kind >= 0:
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        assert lib.probe_launch(kind, table.data_ptr(), table.numel(), out.data_ptr(), usec, blocks, side.cuda_stream) == 0
    h.backward(gout)

names = {-1: "none", 0: "alu", 1: "gather cached", 2: "gather device-scope", 3: "atomics"}
for rep in range(2):
    for kind in (-1, 0, 1, 2, 3, -1):
        for _ in range(300):
            step(kind)
        v = []
        for _ in range(8):
            torch.cuda.synchronize()
            a = time.perf_counter()
            for _ in range(200):
                step(kind)
            torch.cuda.synchronize()
            v.append((time.perf_counter() - a) / 200 * 1e3)
        v.sort()
        print(f"{names[kind]:20s} ({blocks} x 256 threads, {usec:.0f} us beside the backward): min {v[0]:.4f} median {v[len(v) // 2]:.4f} ms/step", flush=True)
