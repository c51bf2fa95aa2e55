"""Drives tools/lab/spmm_lab.hip on the S-arxiv graph: correctness vs the product SpMM, HIP-event timing."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import gconv_adapter_amd as ga

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "spmm_lab.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "spmm_lab.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-Wno-unused-value", "-o", so, os.path.join(HERE, "spmm_lab.hip")], check=True)
if not torch.cuda.is_available():
    print("built", so); sys.exit(0)
lab = C.CDLL(so)
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
batch, cfg = ga.synth.make_workload(name)
n, r = batch.num_nodes, 32
plan = ga.get_plan(batch.edge_index.to(dev), n, True)
nnz = plan.fwd.nnz
h = torch.randn(n, r, device=dev)
lib = ga._lib.load()
ws = torch.empty(max(int(lib.gconv_spmm_workspace_bytes(plan.fwd.ref(), r)), 16), dtype=torch.uint8, device=dev)
fl = ga._lib.SPMM_IN_PRESCALED | ga._lib.SPMM_OUT_PRESCALE
ref = torch.empty_like(h)
ga.spmm(plan, h, flags=fl, out=ref, workspace=ws)
rowptr, col, dis = plan.fwd.rowptr, plan.fwd.col, plan.dis
deg = (rowptr[1:] - rowptr[:-1])
short = (deg <= 64)
stream = torch.cuda.current_stream().cuda_stream
algo = 4 * (n + 1) + 4 * nnz + 4 * n + 8 * n * r


def ev(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def check(y, what):
    ok = torch.allclose(y[short], ref[short], rtol=1e-5, atol=1e-5)
    return "ok" if ok else "MISMATCH %.3e" % (y[short] - ref[short]).abs().max().item()


t = ev(lambda: ga.spmm(plan, h, flags=fl, out=ref, workspace=ws))
print(f"product           {t:6.2f} us  {algo/t/1e3:6.0f} GB/s  (n_long {plan.fwd.n_long} n_seg {plan.fwd.n_seg})", flush=True)
for remap in (0, 1):
    y = torch.zeros_like(h)
    f = lambda: lab.lab_v1(C.c_void_p(rowptr.data_ptr()), C.c_void_p(col.data_ptr()), C.c_void_p(dis.data_ptr()),
                           C.c_void_p(h.data_ptr()), C.c_void_p(y.data_ptr()), n, remap, C.c_void_p(stream))
    assert f() == 0
    t = ev(f)
    print(f"v1 remap={remap}        {t:6.2f} us  {algo/t/1e3:6.0f} GB/s  {check(y, 'v1')}", flush=True)

# ---- row schedules: which XCD handles which rows (blocks are dealt round-robin over the 8 XCDs)
import numpy as np
ei = batch.edge_index.numpy(); src, dst = ei[0], ei[1]
keep = src != dst; src, dst = src[keep], dst[keep]
degn = np.bincount(dst, minlength=n)


def distinct(xcd_of_row):
    key = np.unique(xcd_of_row[dst].astype(np.int64) * n + src)
    selfk = xcd_of_row.astype(np.int64) * n + np.arange(n)
    return len(np.union1d(key, selfk))


def sched_from_assign(assign):
    """slot order such that block b (32 slots) holds rows of XCD b % 8; rows of an XCD keep index order"""
    lists = [np.flatnonzero(assign == x) for x in range(8)]
    nb = (n + 31) // 32
    out = np.full(nb * 32, -1, dtype=np.int64)
    ptr = [0] * 8
    spill = []
    for b in range(nb):
        x = b % 8
        take = lists[x][ptr[x]:ptr[x] + 32]
        ptr[x] += len(take)
        out[b * 32:b * 32 + len(take)] = take
    rest = np.concatenate([lists[x][ptr[x]:] for x in range(8)])
    holes = np.flatnonzero(out < 0)[:len(rest)]
    out[holes] = rest
    assert (np.sort(out[out >= 0]) == np.arange(n)).all()
    return out[:n] if (out[:n] >= 0).all() else out


def run_sched(name, sched_np):
    sched = torch.from_numpy(sched_np.astype(np.int32)).to(dev)
    y = torch.zeros_like(h)
    f = lambda: lab.lab_v1s(C.c_void_p(rowptr.data_ptr()), C.c_void_p(col.data_ptr()), C.c_void_p(dis.data_ptr()),
                            C.c_void_p(h.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(sched.data_ptr()),
                            int(sched.numel()), C.c_void_p(stream))
    assert f() == 0
    t = ev(f)
    print(f"v1s {name:28s} {t:6.2f} us  {algo/t/1e3:6.0f} GB/s  {check(y, 'v1s')}", flush=True)


ident = np.arange(n)
print("distinct (xcd,row) pairs, identity schedule:", distinct((ident // 32) % 8))
run_sched("identity", ident)
rng = np.random.default_rng(0)
assign = (ident // 32) % 8
w = degn + 1
for it in range(12):
    cnt = np.zeros((n, 8), dtype=np.int32)
    np.add.at(cnt, (dst, assign[src]), 1)
    load = np.bincount(assign, weights=w, minlength=8)
    pen = (load / load.mean() - 1) * 6
    new = np.argmax(cnt - pen[None, :], axis=1)
    mv = rng.random(n) < 0.5
    assign = np.where(mv, new, assign)
    if it in (2, 5, 11):
        load = np.bincount(assign, weights=w, minlength=8)
        print(f"label-prop it {it}: distinct {distinct(assign)}  load max/mean {load.max()/load.mean():.3f}  rows max/mean {np.bincount(assign, minlength=8).max()/ (n/8):.3f}")
        run_sched(f"label-prop it{it}", sched_from_assign(assign))
# degree-sorted inside the schedule (heavy rows first) as a control
run_sched("degree-desc", np.argsort(-degn, kind="stable"))
sys.exit(0)
for S in (4, 8, 16):
    for K in (1, 8, 16, 32):
        for remap in (0, 1):
            y = torch.zeros_like(h)
            f = lambda: lab.lab_v2(C.c_void_p(rowptr.data_ptr()), C.c_void_p(col.data_ptr()), C.c_void_p(dis.data_ptr()),
                                   C.c_void_p(h.data_ptr()), C.c_void_p(y.data_ptr()), n, nnz, S, K, remap, C.c_void_p(stream))
            rc = f()
            if rc != 0:
                print("v2", S, K, "launch rc", rc); continue
            t = ev(f, iters=30, warm=5)
            print(f"v2 S={S:2d} K={K:2d} remap={remap} {t:6.2f} us  {algo/t/1e3:6.0f} GB/s  {check(y, 'v2')}", flush=True)
