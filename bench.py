#!/usr/bin/env python3
"""GConv-Adapter hot-path benchmark (contract: see the task's bench.py section).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hiv|esol|...] [--scaling strong|weak]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher: before
torch (or anything else that could touch the GPU) is imported it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 … bench.py <same flags>`
as a CHILD process (one rank per GPU), relays the ranks' output, prints rank 0's JSON line as its own
last stdout line and exits with the child's return code. Under `torch.distributed.run` (WORLD_SIZE set)
the process is a rank. The reference has no counterpart: it is single-process, `cuda:0`
(`/root/reference/scripts/finetune_inductive_learning.py:179`).

One "step" = one pass of the adapter hot path over one collated batch, exactly
what the adapters of one model forward+backward cost in the reference's train
loop (`/root/reference/scripts/finetune_inductive_learning.py:43-67`):

    graph prep for the (new) batch's edge_index      (K0, shared by all adapters)
    A adapters forward, chained                       (A = adapters per model forward)
    backward through all of them (dX + parameter grads)
    [N > 1] one flat grad all-reduce (RCCL over xGMI; gloo only when asked for by GCONV_BENCH_BACKEND)

through the product surface (`GConvAdapter` nn.Module -> autograd.Function ->
C ABI -> HIP kernels). Inputs are resident in HBM before the timed region.

Default workload: BASELINE.json configs[2] — the 4096-molecule HIV-like batch (d=300, r=8, 5
pre-adapters) cut BY GRAPH into N shards, one per GPU (`--scaling strong`, the default for molecule
workloads: total work fixed, N = 1 runs the whole batch on one GPU, N = 8 runs 512 molecules per
GPU — north_star's "≥ 6x at 8 GPUs on sharded HIV batches" is value(N=8) / value(N=1)).
`--scaling weak`: every rank runs its own 512-molecule shard (the round 1-3 headline; at N = 1 it is
reported under extra.hiv512_*). `--workload esol` is configs[1]; both molecule workloads are
reported at N=1 (the other one under "extra").

Timing: an untimed pre-roll (`settle`, reported as config.settle_steps) runs the step until
its rate is steady — a fresh process on the GPU box needs 0.1–0.7 s of activity before the
(host-bound) step reaches its sustained rate, longer than a 20-step warm-up — then W warm-up
steps, then EXACTLY K timed steps between barrier + synchronize brackets.

The JSON line also carries
  roofline      the normalized-adjacency SpMM at S-arxiv scale (north_star's
                roofline subject), timed live with HIP events on the launch stream
  cpu_baseline  the oracle (PyG op order, fp32, autograd) on the host cores,
                same workload, bounded sample            (rank 0, N=1 only)

`--graph-replay` (started by the default run as a child process): configs[3]'s adapter step on the static Cora / PubMed
graphs, eager against ONE replayed HIP graph — one JSON line, merged into `extra`.
"""
from __future__ import annotations

import argparse
from typing import Optional
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# --------------------------------------------------------------------------- launcher (stdlib only)
def _requested_gpus(argv) -> int:
    """`--gpus N` / `--gpus=N` out of an argument list (1 when absent or unparsable)."""
    n = 1
    for i, a in enumerate(argv):
        try:
            if a == "--gpus" and i + 1 < len(argv):
                n = int(argv[i + 1])
            elif a.startswith("--gpus="):
                n = int(a.split("=", 1)[1])
        except ValueError:
            pass
    return n


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(argv, n: int, port: int) -> list:
    """The child command of the launcher: the driver's own documented form for N > 1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + list(argv)


def _visible_gpus() -> int:
    """`torch.cuda.device_count()` asked of a short-lived CHILD process, so that the launcher itself
    never loads torch or the HIP runtime (-1 when the question cannot be answered)."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return -1


def launch_ranks(argv) -> int:
    """Launcher role (see the module docstring). Returns the exit code for this process.
    GCONV_BENCH_LAUNCH_DRYRUN=1 prints what would be started (and whether torch is loaded) instead."""
    import subprocess
    n = _requested_gpus(argv)
    cmd = launcher_command(argv, n, _free_port())
    if os.environ.get("GCONV_BENCH_LAUNCH_DRYRUN"):
        print(json.dumps({"launcher_cmd": cmd, "torch_loaded": "torch" in sys.modules,
                          "package_loaded": "gconv_adapter_amd" in sys.modules}))
        return 0
    have = _visible_gpus()
    if 0 <= have < n and os.environ.get("GCONV_BENCH_BACKEND") != "gloo":
        print(f"bench.py: --gpus {n} but {have} GPU(s) visible; set GCONV_BENCH_BACKEND=gloo for a "
              f"functional rehearsal with several ranks per GPU (never a measurement)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1, env=env)
    line = None
    for out in proc.stdout:
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s            # rank 0's result: held back, printed last
        else:
            sys.stdout.write(out)
            sys.stdout.flush()
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        rc = 1
    return rc


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _requested_gpus(sys.argv[1:]) > 1:
    sys.exit(launch_ranks(sys.argv[1:]))      # nothing below has run: no torch, no HIP in this process

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# the frozen layers' nn.Linear (fine-tune step in the extras): this torch build's hipBLASLt
# addmm path blocks the host at every call until the GPU has caught up (the step ran at GPU time +
# host time, 3.3 ms); with rocBLAS the host runs ahead again (2.4 ms). Read at the first addmm.
os.environ.setdefault("DISABLE_ADDMM_HIP_LT", "1")
import gconv_adapter_amd as ga  # noqa: E402

RESIDENT = 0
PREFETCH_AT = os.environ.get("GCONV_BENCH_PREFETCH_AT", "auto")   # where the step starts the next plans' build
PARTITION = "graphs"
STRONG_WORKLOAD = "hiv4096"   # --scaling strong: the whole batch of BASELINE configs[2], split over the ranks
_FULL_AFFINITY = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
_PINNED = []


def pin_host_threads() -> None:
    """GPU-driving phases: all host threads on a few cores of the GPU's NUMA node
    (gconv_adapter_amd.dist.pin_to_local_cores — one process per GPU, disjoint core sets)."""
    global _PINNED
    _PINNED = ga.dist.pin_to_local_cores()


def unpin_host_threads() -> None:
    """CPU baseline: the oracle gets every core the process is entitled to."""
    if _FULL_AFFINITY is not None and _PINNED:
        ga.dist.set_process_affinity(_FULL_AFFINITY)


HBM_PEAK_GBS = 8000.0       # MI355X spec (MI355X_MICROARCH.md chip table)
HBM_MEASURED_GBS = 6290.0   # float4-copy achievable, same table
FP32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_* dense peak (= fp32 vector peak), same table
MALL_GATHER_GBS = 8600.0    # uniformly random rows of a 38 MB table out of the Infinity Cache (same guide, "Indexed rows")
SPMM_ARXIV_FLOOR_BYTES = 130_900_000    # S-arxiv, rank 32: what must cross the fabric with eight private L2s (LAB.md)
SPMM_GATHER_SUSTAINED_GBS = 6400.0      # what the SpMM's gathers sustain in real traffic (tools/gather_probe, PMC)


# --------------------------------------------------------------------------- helpers
def build_adapters(cfg: dict, normalize: bool, device, cls=None) -> torch.nn.ModuleList:
    cls = cls or ga.GConvAdapter
    mods = torch.nn.ModuleList([
        cls(cfg["d"], cfg["r"], learnable_scalar=True, skip_connection=True,
            normalize=normalize) for _ in range(cfg["adapters"])]).to(device)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():  # trained-scale weights (init scale would be a near-identity)
        for p in mods.parameters():
            p.copy_((torch.randn(p.shape, generator=g) * 0.1).to(device))
        for m in mods:
            m.scalar.fill_(1.0)
    return mods


class BatchStream:
    """K distinct collated batches resident in HBM, served round-robin: what the
    reference's DataLoader hands to the train loop. Every step builds the graph plan of its
    batch anew (the plan is discarded after the step); the build of the batch two steps ahead is
    started before the current step's adapters run, so it overlaps them — the role the
    DataLoader's prefetching workers play for collation in the reference."""

    def __init__(self, name: str, rank: int, device, k: int, fresh: bool, prefetch: bool = True,
                 shard=None, depth: int = 2, rotate: bool = False):
        self.fresh = fresh
        self.prefetch = prefetch   # False: the plan is built in-stream by the step's first adapter
        # plans are built `depth` batches ahead — torch's DataLoader keeps prefetch_factor = 2 batches per
        # worker in flight for the same reason: one batch ahead, a build that takes 0.7 of a step next to the
        # step's own kernels misses its deadline every few hundred steps and the host then waits for it
        # (tools/lab/settle_trace.py: a floor of 0.92 ms with seconds-long stretches at 0.95 - 1.2 at depth 1,
        # 0.925 +- 0.001 at depth 2; GCONV_BENCH_PREFETCH_DEPTH=1 for the A/B)
        self.depth = max(1, min(int(os.environ.get("GCONV_BENCH_PREFETCH_DEPTH", depth)), k - 1)) if fresh else 1
        self.items = []
        cfg = None
        for i in range(k if fresh or rotate else 1):   # (rotate: lab — k batches in turn, their plans kept)
            if shard is None:
                batch, cfg = ga.synth.make_workload(name, seed=rank * 1000 + i)
            else:                  # strong scaling: every rank cuts ITS graphs out of the same batch
                whole, cfg = ga.synth.make_workload(name, seed=i)
                batch = ga.synth.shard_graphs(whole, *shard)
            x = batch.x.to(device)
            ei = batch.edge_index.to(device)
            gout = torch.randn(x.shape, generator=torch.Generator().manual_seed(2 + i)).to(device)
            self.items.append((batch, x, ei, gout))
        self.cfg = cfg
        self.pos = 0
        # where the step starts the build of the plan `depth` batches ahead (start_prefetch): between forward and
        # backward for batches that fill the chip (>= 4096 nodes), at the top of the step for the reference's own
        # 32-molecule batches (host-bound: a call in the middle of the step only delays the backward's launches —
        # ESOL 0.154 -> 0.159 ms). GCONV_BENCH_PREFETCH_AT=top|backward forces one (A/B).
        nodes = max(int(b.num_nodes) for b, *_ in self.items)
        self.prefetch_beside_backward = (PREFETCH_AT == "backward") if PREFETCH_AT in ("top", "backward") else nodes >= 4096
        self.ready_event = (ga.CompletionEvent() if self.prefetch_beside_backward and self.fresh
                            and not os.environ.get("GCONV_NO_COMPLETION_EVENT") else None)

    def avg(self, attr) -> float:
        return sum(float(getattr(b, attr)) for b, *_ in self.items) / len(self.items)

    def next(self, normalize: bool):
        batch, x, ei, gout = self.items[self.pos % len(self.items)]
        if self.pos == 0 or not self.prefetch_beside_backward:
            self.start_prefetch(normalize, self.pos)
        self.pos += 1
        return batch, x, ei, gout

    def start_prefetch(self, normalize: bool, step: Optional[int] = None, ready=None) -> None:
        """Starts the graph build of the batch `depth` steps ahead (all of the first `depth` batches before the
        first step). The build's side stream waits for what the main stream holds at THIS point
        (`prefetch_plan`: an event recorded at submit time), so WHERE the loop calls it decides what the build
        runs beside on the GPU: `adapter_step` calls it between the forward and the backward pass. Beside the
        forward kernels — 10 waves x 168 VGPRs per CU: no build wave fits next to them, every CU a build
        workgroup holds is lost to the forward — the whole-batch step paid 100 us (forward kernels 62 -> 82 us
        each, tools/lab/r05_k0trace.sh); the backward kernels (8 waves x 203 VGPRs) leave room and do not slow down.
        GCONV_BENCH_PREFETCH_AT=top: at the top of the step, as in rounds 1-4 (A/B). `ready`: a `CompletionEvent` the
        forward pass's last adapter call has been armed with — the build waits for that kernel's own completion signal
        and nothing is queued on the main stream (GCONV_NO_COMPLETION_EVENT=1: the recorded event, A/B)."""
        if not (self.fresh and self.prefetch):
            return
        if step is None:
            step = self.pos - 1                           # (the step that is running: next() has counted it)
        ahead = range(1, self.depth + 1) if step == 0 else (self.depth,)
        for a in ahead:
            _, nx, nei, _ = self.items[(step + a) % len(self.items)]
            ga.prefetch_plan(nei, nx.shape[-2], normalize, ready=ready)


class StoreStream:
    """`--resident`: the whole dataset lives in HBM (`MoleculeStore`); every step draws a fresh
    random batch, collates it on the device and assembles its graph plan from the dataset's plan
    (no sort, no host synchronisation) — an epoch of the reference's shuffled DataLoader without
    the CPU collation and the H2D copies."""
    fresh = False   # (plans are registered by the store; nothing to prefetch or discard)

    def __init__(self, name: str, rank: int, device, dataset_graphs: int):
        import numpy as np
        cfg = dict(ga.synth.WORKLOADS[name])
        whole = ga.synth.molecule_batch(cfg["mol"], dataset_graphs, hidden=cfg["d"], seed=rank * 1000)
        self.store = ga.MoleculeStore.from_batch(whole, device)
        self.cfg, self.per_batch = cfg, cfg["graphs"]
        self.rng = np.random.default_rng(rank)
        biggest = int((whole.graph_ptr[1:] - whole.graph_ptr[:-1]).sort().values[-self.per_batch:].sum())
        self.gout_all = torch.randn(biggest, cfg["d"], generator=torch.Generator().manual_seed(2)).to(device)
        self.nodes = self.edges = self.count = 0
        self.store._dataset_plan(True); self.store._dataset_plan(False)

    def avg(self, attr) -> float:
        return {"num_graphs": self.per_batch, "num_nodes": self.nodes / max(self.count, 1)}[attr]

    def next(self, normalize: bool):
        ids = self.rng.choice(self.store.num_graphs, self.per_batch, replace=False)
        batch = self.store.collate(ids, normalize=normalize)
        self.nodes += batch.num_nodes; self.edges += batch.edge_index.shape[1]; self.count += 1
        return batch, batch.x, batch.edge_index, self.gout_all[:batch.num_nodes]


def adapter_step(mods, params, stream: BatchStream, normalize: bool, bucket=None):
    """fwd + bwd of the adapter chain on the next batch (+ gradient all-reduce)."""
    batch, x, ei, gout = stream.next(normalize)
    if bucket is not None:
        bucket.wait()         # the previous step's all-reduce, before its buffer is written again
    ga.optim.zero_grad(params)     # optimizer.zero_grad(set_to_none=True): one host call for the parameter list
    beside = isinstance(stream, BatchStream) and stream.prefetch_beside_backward and stream.pos > 1
    h = x
    for m in mods[:-1]:
        h = m(h, ei)
    if beside and stream.ready_event is not None:
        stream.ready_event.arm()       # the last forward kernel's completion is what the build waits for
    h = mods[-1](h, ei)
    if beside:
        stream.start_prefetch(normalize, ready=stream.ready_event)   # the next batches' graph build runs beside the BACKWARD pass
    h.backward(gout)
    if bucket is not None:
        # the backward pass has written every gradient into the flat bucket (gradient sinks):
        # ONE collective, launched behind the backward, waited for at the top of the next step
        bucket.all_reduce_grads(async_op=True)
    if (stream.fresh or isinstance(stream, StoreStream)) and not os.environ.get("GCONV_LAB_NO_DISCARD"):
        ga.PLAN_CACHE.discard(ei, x.shape[-2], normalize)   # a training batch is seen once
    return batch.num_graphs


def settle(fn, device, world: int = 1, chunk: int = 100, min_seconds: float = 1.5,
           max_seconds: float = 4.0) -> int:
    """Untimed pre-roll until the step rate is steady. A fresh process on the GPU box runs the
    same step 1.3–1.8x slower for its first 0.1–0.7 s (hundreds to ~2000 steps; host-side
    ramp-up of the box: the step is host-bound, back-to-back kernel time does not change) —
    longer than any sensible --warmup. Runs chunks of `chunk` steps for at least `min_seconds`,
    until two consecutive chunks agree within 3 % (or `max_seconds`); returns the number of steps."""
    if world > 1:   # the step contains a collective: every rank must run the same number of steps,
        # so the counts come from max-reduced timings: ~2 s of steps, at most 40 chunks
        ran, spent, n = 0, 0.0, 5
        while True:
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(device)
            took = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            dist.all_reduce(took, op=dist.ReduceOp.MAX)
            took = max(float(took.item()), 1e-6)
            ran, spent = ran + n, spent + took
            if spent >= 2.0 or ran >= 40 * chunk:
                return ran
            n = int(min(40 * chunk - ran, max(5, (2.0 - spent) / (took / n))))
    ran, prev = 0, None
    t_begin = time.perf_counter()
    t_end = t_begin + max_seconds
    while time.perf_counter() < t_end:
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(chunk):
            fn()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        ran += chunk
        if prev is not None and abs(dt - prev) <= 0.03 * prev and \
                time.perf_counter() - t_begin >= min_seconds:   # (plateaus of ~0.7 s were seen)
            break
        prev = dt
    return ran


def timed_steps(fn, steps: int, warmup: int, world: int, device) -> float:
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(device)
    own = time.perf_counter() - t0      # this rank's own K steps (before it waits for the others)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, own / steps * 1e3)
        LAST_DIAG["ms_per_step_per_rank"] = [round(v, 4) for v in per_rank]
    return dt


LAST_DIAG: dict = {}   # what the last timed_steps / collective probe of an N > 1 run saw (rank 0 prints it)


def probe_all_reduce(bucket, world: int, device, samples: int = 10) -> dict:
    """The step's ONE collective on its own, OUTSIDE the timed region (N > 1 only): `samples` all-reduces of
    the flat gradient bucket, each behind a barrier (arrival skew is not the collective's time), timed with HIP
    events on the current stream (RCCL: the events bracket the stream's wait for the collective — its device
    time; gloo stages through the host: only the wall figure means anything) and with the host clock."""
    ev, wall = [], []
    for _ in range(samples + 2):
        torch.cuda.synchronize(device)
        dist.barrier()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        bucket.all_reduce(async_op=False)
        b.record()
        b.synchronize()
        wall.append((time.perf_counter() - t0) * 1e6)
        ev.append(a.elapsed_time(b) * 1e3)
    ev, wall = sorted(ev[2:]), sorted(wall[2:])          # (the first two: connection set-up, clocks)
    mine = {"event_us_median": round(ev[len(ev) // 2], 1), "event_us_min": round(ev[0], 1),
            "wall_us_median": round(wall[len(wall) // 2], 1), "bytes": bucket.nbytes}
    every = [None] * world
    dist.all_gather_object(every, mine)
    return {"bytes": bucket.nbytes, "samples": samples,
            "event_us_median_max_over_ranks": max(e["event_us_median"] for e in every),
            "event_us_min_min_over_ranks": min(e["event_us_min"] for e in every),
            "wall_us_median_max_over_ranks": max(e["wall_us_median"] for e in every)}


def event_time_ms(fn, iters: int, warmup: int = 5) -> float:
    """Average duration of `fn` (kernels on torch's current stream) via HIP events."""
    for _ in range(warmup):
        fn()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters


def best_event_time_ms(fn, iters: int, reps: int = 3, warmup: int = 5) -> float:
    """Best of `reps` averages: the small extras are launch-bound and one host hiccup (allocator,
    GC, a scheduler move) in 50 iterations doubles an average."""
    return min(event_time_ms(fn, iters, warmup) for _ in range(reps))


# --------------------------------------------------------------------------- roofline
def spmm_algorithmic_bytes(n: int, nnz: int, f: int) -> int:
    """rowptr + col + dis + read H + write Y, each touched once."""
    return 4 * (n + 1) + 4 * nnz + 4 * n + 4 * n * f + 4 * n * f


def adapter_algorithmic_bytes(n: int, nnz: int, d: int, r: int):
    idx = 4 * (n + 1) + 4 * nnz + 4 * n
    fwd = 2 * 4 * n * d + 6 * 4 * n * r + 2 * idx
    bwd = 3 * 4 * n * d + 8 * 4 * n * r + 2 * idx
    return fwd, bwd


def measure_arxiv(device, iters: int = 50, banded_control: bool = False) -> dict:
    """North-star roofline subject: D^-1/2 (A+I) D^-1/2 SpMM at rank 32 on S-arxiv,
    plus one whole adapter fwd / bwd at the same scale."""
    batch, cfg = ga.synth.make_workload("arxiv")
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    ei = batch.edge_index.to(device)
    plan = ga.get_plan(ei, n, True)
    nnz = plan.fwd.nnz
    h = torch.randn(n, r, device=device)
    y = torch.empty_like(h)
    flags = ga._lib.SPMM_IN_PRESCALED | ga._lib.SPMM_OUT_PRESCALE  # the variant the adapter runs
    lib = ga._lib.load()
    ws = torch.empty(max(int(lib.gconv_spmm_workspace_bytes(plan.fwd.ref(), r)), 16),
                     dtype=torch.uint8, device=device)
    ms = event_time_ms(lambda: ga.spmm(plan, h, flags=flags, out=y, workspace=ws), iters)
    b = spmm_algorithmic_bytes(n, nnz, r)
    achieved = b / (ms * 1e-3) / 1e9
    # ONE launch per SpMM: spmm_rows_kernel<32,true,0> (rows, long-row segments and their
    # completion); avg_launch_us is its rocprofv3 average
    roof = {"bound": "hbm", "kernel": "spmm_rows_kernel<32,norm,generic>",
            "workload": f"S-arxiv N={n} nnz={nnz} F={r} normalized",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "frac_of_measured_copy_peak": round(achieved / HBM_MEASURED_GBS, 4),
            "algorithmic_bytes": b, "avg_launch_us": round(ms * 1e3, 2), "traffic": None}
    pmc = next((q for q in (os.path.join(ROOT, "profiles", f"r0{k}_spmm_arxiv_pmc.json") for k in (5, 4, 3, 2, 1))
                if os.path.exists(q)), "")
    if os.path.exists(pmc):   # PMC passes cannot run inside the timed bench: committed rocprofv3 result
        with open(pmc) as fh:
            dv = json.load(fh)["derived"]
        roof["traffic"] = round(dv["traffic_bytes_per_launch"])
        # What the counters count: bytes between the eight L2s and the FABRIC per launch. FETCH_SIZE includes
        # Infinity-Cache hits (MI355X_MICROARCH.md, HBM), and the 21.8 MB H table is MALL-resident — so this is
        # fabric / MALL traffic, not HBM traffic, and its yardstick is the guide's rate for uniformly random
        # rows out of the Infinity Cache (8.6 TB/s, "Indexed rows"), not the 6.29 TB/s HBM copy peak.
        roof["traffic_kind"] = "L2<->fabric bytes incl. Infinity-Cache hits (H table MALL-resident): not HBM bytes"
        roof["traffic_GBs"] = round(dv["traffic_bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
        roof["traffic_frac_of_mall_gather_rate"] = round(roof["traffic_GBs"] / MALL_GATHER_GBS, 3)
        roof["traffic_source"] = (f"profiles/{os.path.basename(pmc)} (collected at commit "
                                  f"{dv.get('commit', 'n/a')}): 2 x FETCH_SIZE + WRITE_SIZE per launch "
                                  "(separate rocprofv3 --pmc passes, gfx950 x2 fetch correction); L2 hit rate "
                                  f"{dv['l2_hit_rate']:.2f}")
    # The structural floor of a row-partitioned gather on THIS graph (seeded generator; LAB.md "SpMM floor"):
    # the rows an XCD processes reference 802,599 distinct (XCD, H-row) pairs -> 102.7 MB must cross the fabric even
    # with eight infinite L2s (21.8 MB with one), + 21.9 MB of output + 6.3 MB of index data = 130.9 MB, at the
    # 6.4 TB/s the gathers sustain (tools/gather_probe: 1.12 x below the kernel). north_star's 0.50 is below it.
    roof["floor_bytes"] = SPMM_ARXIV_FLOOR_BYTES
    roof["floor_us"] = round(SPMM_ARXIV_FLOOR_BYTES / (SPMM_GATHER_SUSTAINED_GBS * 1e9) * 1e6, 1)
    roof["floor_frac"] = round(b / (roof["floor_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)
    roof["frac_of_floor"] = round(roof["floor_us"] / (ms * 1e3), 3)
    roof["target_frac"] = 0.5
    roof["target_met"] = bool(roof["frac"] >= 0.5)

    banded = {}
    if banded_control:
        banded = measure_banded(device, r, flags, lib)

    mod = build_adapters(dict(d=d, r=r, adapters=1), True, device)[0]
    x = batch.x.to(device).requires_grad_(True)
    gout = torch.randn(n, d, device=device)
    out = mod(x, ei)
    fwd_ms = event_time_ms(lambda: mod(x, ei), 20)
    leaves = [x] + list(mod.parameters())

    def bwd():   # fresh .grad every time: no accumulation kernels inside the timed region
        for q in leaves:
            q.grad = None
        out.backward(gout, retain_graph=True)
    bwd_ms = event_time_ms(bwd, 20)
    fb, bb = adapter_algorithmic_bytes(n, nnz, d, r)
    extra = {"arxiv_adapter_fwd_us": round(fwd_ms * 1e3, 1), "arxiv_adapter_bwd_us": round(bwd_ms * 1e3, 1),
             "arxiv_adapter_fwd_GBs": round(fb / (fwd_ms * 1e-3) / 1e9, 1),
             "arxiv_adapter_bwd_GBs": round(bb / (bwd_ms * 1e-3) / 1e9, 1),
             "arxiv_adapter_steps_per_s": round(1e3 / (fwd_ms + bwd_ms), 1)}
    with torch.no_grad():   # the evaluation pass (finetune_transductive_learning.py:233-260): forward only, nothing saved
        extra["arxiv_adapter_eval_fwd_us"] = round(event_time_ms(lambda: mod(x, ei), 20) * 1e3, 1)
    del out
    # BASELINE configs[4] as ONE step: L = 2 layers x (pre + post) = 4 adapters on the static graph, ONE shared
    # plan, forward + backward of the chain (what `--workload arxiv` headlines; here next to the default line)
    mods4 = build_adapters(dict(d=d, r=r, adapters=cfg["adapters"]), True, device)
    params4 = list(mods4.parameters())

    def step4():
        for q in params4:
            q.grad = None
        x.grad = None
        h = x
        for m in mods4:
            h = m(h, ei)
        h.backward(gout)
    step4_ms = best_event_time_ms(step4, 10, reps=3, warmup=3)
    extra["arxiv_step4_ms"] = round(step4_ms, 4)
    extra["arxiv_step4_GBs"] = round(cfg["adapters"] * (fb + bb) / (step4_ms * 1e-3) / 1e9, 1)
    extra.update(banded)
    return {"roofline": roof, "extra": extra}


def measure_arxiv_projections(device, iters: int = 50) -> dict:
    """S-arxiv shape, the two projections alone and the graph build (K0)."""
    batch, cfg = ga.synth.make_workload("arxiv")
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    ei = batch.edge_index.to(device)
    plan = ga.get_plan(ei, n, True)
    h = torch.randn(n, r, device=device)
    # the two projections alone (north_star: "MFMA utilisation for the projections against chip peak"):
    # exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), 2 N d Fp flop each; both are HBM-bound by design
    # (arithmetic intensity Fp / 2 flop per byte), so the fraction of the 157.3 TFLOP/s fp32 matrix
    # peak is what a memory-bound skinny GEMM leaves of it
    xs = batch.x.to(device)
    wd = torch.randn(r, d, device=device) * 0.1
    wu = torch.randn(d, r, device=device) * 0.1
    bu = torch.randn(d, device=device)
    sc = torch.ones(1, device=device)
    k1_ms = best_event_time_ms(lambda: ga.gemm_down(xs, wd, plan.dis), iters)
    k3_ms = best_event_time_ms(lambda: ga.gemm_up_fused(h, wu, bu, sc, xs, True), iters)
    flop = 2.0 * n * d * r
    proj = {"arxiv_proj_down_us": round(k1_ms * 1e3, 1), "arxiv_proj_up_us": round(k3_ms * 1e3, 1),
            "arxiv_proj_down_TFLOPs": round(flop / (k1_ms * 1e-3) / 1e12, 1),
            "arxiv_proj_up_TFLOPs": round(flop / (k3_ms * 1e-3) / 1e12, 1),
            "arxiv_proj_down_GBs": round((4 * n * d + 4 * n * r) / (k1_ms * 1e-3) / 1e9, 1),
            "arxiv_proj_up_GBs": round((8 * n * d + 4 * n * r) / (k3_ms * 1e-3) / 1e9, 1),
            "arxiv_proj_frac_of_fp32_mfma_peak": round(2 * flop / ((k1_ms + k3_ms) * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3)}
    del xs
    # K0 ("prep reported separately and amortized"): one graph build from the COO list,
    # shared by the 4 adapters x (2 fwd + 2 bwd) propagations of a step on this static graph
    def build_plan():
        ga.GraphPlan(ei, n, True)
    k0_ms = event_time_ms(build_plan, 10, warmup=2)
    proj["arxiv_graph_build_us"] = round(k0_ms * 1e3, 1)

    ga.PLAN_CACHE.clear()
    return proj


def measure_tile_kernels(device, name: str = "hiv4096") -> dict:
    """Device time of ONE adapter forward / backward on the whole 4096-molecule batch of
    configs[2] resident on one GPU (fused tile kernels through the module, HIP events), at
    algorithmic bytes: what the kernels sustain when a workgroup has a dozen tiles
    to walk instead of one or two (the 512-molecule shard is bounded by its largest tile)."""
    batch, cfg = ga.synth.make_workload(name)
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    ei = batch.edge_index.to(device)
    plan = ga.get_plan(ei, n, True)
    mod = build_adapters(dict(d=d, r=r, adapters=1), True, device)[0]
    x = batch.x.to(device).requires_grad_(True)
    gout = torch.randn(n, d, device=device)
    out = mod(x, ei)
    fwd_ms = event_time_ms(lambda: mod(x, ei), 30)
    leaves = [x] + list(mod.parameters())

    def bwd():
        for q in leaves:
            q.grad = None
        out.backward(gout, retain_graph=True)
    bwd_ms = event_time_ms(bwd, 30)
    with torch.no_grad():   # the reference's evaluation passes (finetune_inductive_learning.py:95-104): forward only
        eval_ms = event_time_ms(lambda: mod(x, ei), 30)
    fb, bb = adapter_algorithmic_bytes(n, plan.fwd.nnz, d, r)
    ga.PLAN_CACHE.clear()
    res = {f"{name}_adapter_fwd_us": round(fwd_ms * 1e3, 1), f"{name}_adapter_bwd_us": round(bwd_ms * 1e3, 1),
           f"{name}_adapter_eval_fwd_us": round(eval_ms * 1e3, 1),
           f"{name}_adapter_fwd_GBs": round(fb / (fwd_ms * 1e-3) / 1e9, 1),
           f"{name}_adapter_bwd_GBs": round(bb / (bwd_ms * 1e-3) / 1e9, 1),
           f"{name}_adapter_frac_of_hbm_peak": round((fb + bb) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
    pmc = next((q for q in (os.path.join(ROOT, "profiles", f"r0{k}_tile_traffic_{name}.json") for k in (5, 4)) if os.path.exists(q)), "")
    if os.path.exists(pmc):   # what the two kernels REALLY move per launch (committed rocprofv3 --pmc passes), at the live times
        with open(pmc) as fh:
            tt = json.load(fh)
        tf, tb = tt["tile_fwd"]["traffic_bytes_per_launch"], tt["tile_bwd"]["traffic_bytes_per_launch"]
        res[f"{name}_adapter_traffic_MB"] = [round(tf / 1e6, 1), round(tb / 1e6, 1)]
        res[f"{name}_adapter_traffic_frac_of_measured_copy_peak"] = round(
            (tf + tb) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9 / HBM_MEASURED_GBS, 3)
        res[f"{name}_adapter_traffic_source"] = f"profiles/{os.path.basename(pmc)} (commit {tt.get('commit', 'n/a')})"
    return res


def measure_gin(device, name: str = "hiv4096") -> dict:
    """Next to the path: the frozen GIN layer's aggregation (`csrc/gin.hip`) on the
    whole 4096-molecule batch — neighbour sum + bond term forward, d agg -> d x backward — at
    algorithmic bytes (read x, write agg, CSR, [N, 18] code counts), HIP events."""
    batch, cfg = ga.synth.make_workload(name)
    n, d = batch.num_nodes, cfg["d"]
    ei = batch.edge_index.to(device)
    e = ei.shape[1]
    g = torch.Generator().manual_seed(0)
    ea = torch.stack([torch.randint(0, 4, (e,), generator=g), torch.randint(0, 3, (e,), generator=g)], 1).to(device)
    x = batch.x.to(device)
    layer = ga.MolecularGINConv(d).to(device)
    plan = ga.get_plan(ei, n, False)
    counts = ga.gin.edge_code_counts(ei, ea, n)
    t, dr = layer.edge_embedding_type.weight.detach(), layer.edge_embedding_direction.weight.detach()
    fwd_ms = event_time_ms(lambda: ga.gin._aggregate(plan.fwd, x, counts, t, dr), 50)
    bwd_ms = event_time_ms(lambda: ga.gin._aggregate(plan.bwd, x, None, None, None), 50)
    fb = 2 * n * d * 4 + 4 * (n + 1) + 4 * plan.fwd.nnz + 4 * 18 * n
    bb = 2 * n * d * 4 + 4 * (n + 1) + 4 * plan.fwd.nnz
    ga.PLAN_CACHE.clear()
    return {f"{name}_gin_aggregate_fwd_us": round(fwd_ms * 1e3, 1), f"{name}_gin_aggregate_bwd_us": round(bwd_ms * 1e3, 1),
            f"{name}_gin_aggregate_fwd_GBs": round(fb / (fwd_ms * 1e-3) / 1e9, 1),
            f"{name}_gin_aggregate_bwd_GBs": round(bb / (bwd_ms * 1e-3) / 1e9, 1)}


def measure_transductive(device, name: str) -> dict:
    """configs[3]: one GConv-Adapter on a Cora / PubMed-sized graph, x as [1, N, 32], rank 16, normalized
    adjacency with the loops pre-added (`scripts/finetune_transductive_learning.py:111-114`) — the shapes the
    NodeFormer backbone hands the adapter (`nodeformer.py:406-409`). These graphs are launch-bound (a few
    hundred KB per array): the figures are device time per call (HIP events) and the SpMM alone; the
    general path runs K2' + K3 as one gather-fused launch at this size."""
    batch, cfg = ga.synth.make_workload(name)
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    ei = batch.edge_index.to(device)
    plan = ga.get_plan(ei, n, True)
    lib = ga._lib.load()
    fp = int(lib.gconv_padded_rank(r))
    h = torch.randn(n, fp, device=device)
    y = torch.empty_like(h)
    ws = torch.empty(max(int(lib.gconv_spmm_workspace_bytes(plan.fwd.ref(), fp)), 16), dtype=torch.uint8,
                     device=device)
    flags = ga._lib.SPMM_IN_PRESCALED | ga._lib.SPMM_OUT_PRESCALE
    sp_ms = best_event_time_ms(lambda: ga.spmm(plan, h, flags=flags, out=y, workspace=ws), 50)
    mod = build_adapters(dict(d=d, r=r, adapters=1), True, device)[0]
    x = batch.x.reshape(1, n, d).to(device).requires_grad_(True)
    gout = torch.randn(1, n, d, device=device)
    out = mod(x, ei)
    fwd_ms = best_event_time_ms(lambda: mod(x, ei), 50)
    leaves = [x] + list(mod.parameters())

    def bwd():
        for q in leaves:
            q.grad = None
        out.backward(gout, retain_graph=True)
    bwd_ms = best_event_time_ms(bwd, 50)
    ga.PLAN_CACHE.clear()
    return {f"{name}_spmm_us": round(sp_ms * 1e3, 1),
            f"{name}_spmm_GBs": round(spmm_algorithmic_bytes(n, plan.fwd.nnz, fp) / (sp_ms * 1e-3) / 1e9, 1),
            f"{name}_adapter_fwd_us": round(fwd_ms * 1e3, 1), f"{name}_adapter_bwd_us": round(bwd_ms * 1e3, 1)}


def measure_graph_replay(device, name: str) -> dict:
    """configs[3], the transductive loop: the SAME graph every epoch. With the plan built, forward + backward make no
    allocation outside torch's allocator and no host read-back, so the step is captured in ONE HIP graph and replayed
    (tests/test_gpu_parity.py::test_adapter_step_is_hip_graph_capturable: same bits) — what a launch-bound step costs
    without the host in it. Run in a CHILD process by `graph_replay_child` (a failed capture must not cost the line)."""
    batch, cfg = ga.synth.make_workload(name)
    n, d, r = batch.num_nodes, cfg["d"], cfg["r"]
    ei = batch.edge_index.to(device)
    ga.get_plan(ei, n, True)
    mod = build_adapters(dict(d=d, r=r, adapters=1), True, device)[0]
    x = batch.x.reshape(1, n, d).to(device).requires_grad_(True)
    gout = torch.randn(1, n, d, device=device)
    leaves = [x] + list(mod.parameters())

    def whole():
        for q in leaves:
            q.grad = None
        mod(x, ei).backward(gout)
    eager_ms = best_event_time_ms(whole, 50)
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        for _ in range(3):
            whole()
    torch.cuda.current_stream(device).wait_stream(side)
    for q in leaves:
        q.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):   # (no older autograd graph of these leaves is alive here: INTEGRATION.md)
        mod(x, ei).backward(gout)
    replay_ms = best_event_time_ms(graph.replay, 50)
    del graph
    ga.PLAN_CACHE.clear()
    return {f"{name}_adapter_step_eager_us": round(eager_ms * 1e3, 1),
            f"{name}_adapter_step_hip_graph_us": round(replay_ms * 1e3, 1)}


def graph_replay_child(timeout_s: float = 180.0) -> dict:
    """`measure_graph_replay` for Cora and PubMed in a child process (`bench.py --graph-replay`): its last stdout line."""
    if ga._torch_ext.get() is None or os.environ.get("GCONV_BENCH_NO_GRAPH"):
        return {}
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--graph-replay"], capture_output=True, text=True,
                           timeout=timeout_s)
        if r.returncode == 0 and r.stdout.strip():
            return json.loads(r.stdout.strip().splitlines()[-1])
        return {"graph_replay_child": f"failed (rc {r.returncode})"}
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"graph_replay_child": f"failed ({type(e).__name__})"}


def measure_no_prefetch_step(device, normalize: bool, name: str = "esol", steps: int = 200) -> dict:
    """The reference's own batch size with NO change to its train loop: nobody calls `prefetch_plan`,
    every batch's graph plan is built in-stream, synchronously, by the step's first adapter (K0 on the
    critical path: the case the single-workgroup builder is for)."""
    stream = BatchStream(name, 0, device, 4, fresh=True, prefetch=False)
    cfg = stream.cfg
    mods = build_adapters(cfg, normalize, device)
    params = [p for m in mods for p in m.parameters()]

    def step():
        adapter_step(mods, params, stream, normalize, None)
    for _ in range(300):
        step()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(device)
        best = min(best, (time.perf_counter() - t0) / steps)
    ga.PLAN_CACHE.clear()
    return {f"{name}_no_prefetch_ms_per_step": round(best * 1e3, 4),
            f"{name}_no_prefetch_graphs_per_s": round(stream.avg("num_graphs") / best, 1)}


def measure_graph_build(device) -> dict:
    """K0 for molecule batches, in-stream and synchronous (what a caller that does NOT use
    `prefetch_plan` pays in front of a batch's first adapter): wall time per `GraphPlan` for the
    reference's batch size (32 molecules: the single-workgroup builder) and for the 512-molecule shard
    (the multi-launch pipeline)."""
    out = {}
    for name, graphs, key in (("esol", None, "esol32"), ("hiv", None, "hiv512")):
        batch, _ = ga.synth.make_workload(name, graphs=graphs)
        ei = batch.edge_index.to(device)
        n = batch.num_nodes
        for _ in range(5):
            ga.GraphPlan(ei, n, True)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(50):
                ga.GraphPlan(ei, n, True)
            torch.cuda.synchronize(device)
            best = min(best, (time.perf_counter() - t0) / 50)
        out[f"{key}_graph_build_wall_us"] = round(best * 1e6, 1)
    return out


class _FinetuneModel(torch.nn.Module):
    """The reference's inductive model in the shape its fine-tune loop sees it (`MolecularGNN.forward`,
    `src/models/inductive/gnn.py:85-130`, sequential pre-adapters; mean pool; `graph_pred_linear`):
    frozen GIN layers, trainable adapters / BatchNorm / head (`gnn.py:230-241`)."""

    def __init__(self, d: int, r: int, layers: int, tasks: int, normalize: bool):
        super().__init__()
        self.gnn_layers = torch.nn.ModuleList([ga.MolecularGINConv(d) for _ in range(layers)])
        bn = torch.nn.BatchNorm1d if os.environ.get("GCONV_TORCH_BATCH_NORM") else ga.BatchNorm1d   # A/B knob
        self.batch_norm_layers = torch.nn.ModuleList([bn(d) for _ in range(layers)])
        for i, m in enumerate(self.batch_norm_layers):   # the ReLU after the norm (gnn.py:127-128) folded in
            m.fused_relu = isinstance(m, ga.BatchNorm1d) and i != layers - 1
        self.pre_adapters = torch.nn.ModuleDict({str(i): ga.GConvAdapter(d, r, learnable_scalar=True,
                                                                         normalize=normalize) for i in range(layers)})
        self.graph_pred_linear = torch.nn.Linear(d, tasks)
        for layer in self.gnn_layers:
            for p in layer.parameters():
                p.requires_grad_(False)

    def forward(self, x, edge_index, edge_attr, batch, num_graphs, inv_count):
        h = x
        last = len(self.gnn_layers) - 1
        for i, (gin, bn) in enumerate(zip(self.gnn_layers, self.batch_norm_layers)):
            h = self.pre_adapters[str(i)](h, edge_index, edge_attr)
            h = bn(gin(h, edge_index, edge_attr))
            if i != last and not getattr(bn, "fused_relu", False):
                h = torch.relu(h)
        if os.environ.get("GCONV_TORCH_READOUT"):   # A/B knob: torch's scatter instead of csrc/readout.hip
            pooled = torch.zeros(num_graphs, h.shape[1], device=h.device).index_add_(0, batch, h) * inv_count
        else:
            pooled = ga.global_mean_pool(h, batch, num_graphs)          # `self.pool(h, batch)`, gnn.py:187-195
        return self.graph_pred_linear(pooled)


def measure_finetune_step(device, normalize: bool, name: str = "hiv", steps: int = 60) -> dict:
    """The whole fine-tune step of the reference's loop body (`scripts/finetune_inductive_learning.py:43-67`)
    on one batch of the workload (512 molecules: a GPU's share at N = 8; 4096: the whole batch): model forward (5 x adapter -> frozen GIN -> BatchNorm -> ReLU, mean pool,
    head), NaN-masked BCE in fp64, backward, FusedAdam — the adapters in their surroundings, every piece but
    the head's Linear on this package's kernels (~75 launches per step). Reported next to the headline,
    never as `value`."""
    batch, cfg = ga.synth.make_workload(name)
    d, r, n, e = cfg["d"], cfg["r"], batch.num_nodes, batch.edge_index.shape[1]
    g = torch.Generator().manual_seed(0)
    ea = torch.stack([torch.randint(0, 4, (e,), generator=g), torch.randint(0, 3, (e,), generator=g)], 1).to(device)
    sizes = batch.graph_ptr[1:] - batch.graph_ptr[:-1]
    node_graph = torch.repeat_interleave(torch.arange(batch.num_graphs), sizes).to(device)
    inv_count = (1.0 / sizes.clamp(min=1).float()).unsqueeze(1).to(device)
    y = torch.randint(0, 2, (batch.num_graphs, 1), generator=g).double()
    y[torch.rand(y.shape, generator=g) < 0.1] = float("nan")
    y = y.to(device)
    valid = ~torch.isnan(y)
    # the reference indexes with the boolean mask (`output[valid_mask]`, :56-58), which makes the host
    # wait for the forward pass every step; the mask comes with the batch, so its indices are
    # taken once, on the side, and the step stays asynchronous
    valid_idx = valid.view(-1).nonzero().squeeze(1)
    target, count = y[valid], valid.sum()
    model = _FinetuneModel(d, r, cfg["adapters"], 1, normalize).to(device)
    opt = ga.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    x, ei = batch.x.to(device), batch.edge_index.to(device)

    torch_loss = bool(os.environ.get("GCONV_TORCH_READOUT"))

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x, ei, ea, node_graph, batch.num_graphs, inv_count)
        if torch_loss:
            loss = torch.nn.functional.binary_cross_entropy_with_logits(out.view(-1)[valid_idx].double(), target,
                                                                        reduction="none").sum() / count
        else:
            loss = ga.masked_loss(out, y, "bce")     # the loop's NaN-masked fp64 loss (:55-63), one launch
        loss.backward()
        opt.step()

    # best of 3 averages: ~200 host-side dispatches per step, and the first pass after the S-arxiv phases /
    # empty_cache() runs at allocator and clock ramp-up speed (seen: 2.2 - 3.2 ms, then 1.4)
    ms = best_event_time_ms(step, steps, reps=3, warmup=20)
    res = {f"{name}_finetune_step_ms": round(ms, 4),
           f"{name}_finetune_step_graphs_per_s": round(batch.num_graphs / (ms * 1e-3), 1)}
    ga.PLAN_CACHE.clear()
    return res


def measure_banded(device, r: int, flags: int, lib) -> dict:
    """Control (`--banded`): the SAME SpMM kernel on a graph of the same average degree whose
    neighbours are local (banded, 1 M nodes): how close it gets to HBM when gathers hit L2.
    Off by default so that the default run launches the roofline kernel on ONE workload only
    (its rocprofv3 average then equals the live HIP-event figure)."""
    bb = ga.synth.banded_like(1_000_000, 8, 2048)
    bplan = ga.GraphPlan(bb.edge_index.to(device), bb.num_nodes, True)
    bh = torch.randn(bb.num_nodes, r, device=device)
    by = torch.empty_like(bh)
    bws = torch.empty(max(int(lib.gconv_spmm_workspace_bytes(bplan.fwd.ref(), r)), 16),
                      dtype=torch.uint8, device=device)
    bms = event_time_ms(lambda: ga.spmm(bplan, bh, flags=flags, out=by, workspace=bws), 20)
    bbytes = spmm_algorithmic_bytes(bb.num_nodes, bplan.fwd.nnz, r)
    return {"spmm_banded_1M_us": round(bms * 1e3, 1),
            "spmm_banded_1M_GBs": round(bbytes / (bms * 1e-3) / 1e9, 1),
            "spmm_banded_1M_frac_of_hbm_peak": round(bbytes / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


# --------------------------------------------------------------------------- CPU baseline
def host_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribes a container's share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(batch, cfg: dict, normalize: bool, budget_s: float = 20.0) -> dict:
    """The oracle (PyG op order: linear -> gather -> scale -> index_add -> bias, gcn_norm
    recomputed per conv, autograd backward) on the host cores — SURVEY.md 8(d): 3 warm-up steps,
    median of 10 (fewer if 10 do not fit `budget_s`). Also timed: the REASSOCIATED order (both
    propagations at rank width), so that the GPU / CPU ratio can be read without the reference's
    d-wide propagate in it. Checker code timed as the baseline only — it is never on the product path."""
    from oracle import pyg_restated as oracle
    unpin_host_threads()
    threads = host_cores()
    # the baseline's threads on `threads` physical cores of one NUMA node (its best case: 16
    # free-roaming threads under a 16-CPU quota on a 256-thread host ran 2x slower, and erratically)
    cores = ga.dist.local_core_set(0, 1, cores_per_rank=threads)
    if len(cores) == threads and not os.environ.get("GCONV_NO_PIN"):
        ga.dist.set_process_affinity(cores)
    torch.set_num_threads(threads)
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=normalize)
    g = torch.Generator().manual_seed(1)
    params = []
    for _ in range(cfg["adapters"]):
        p = oracle.init_params(cfg["d"], cfg["r"], flags, dtype=torch.float32)
        params.append({k: v.clone().requires_grad_(True) for k, v in p.items()})
    x, ei = batch.x, batch.edge_index
    gout = torch.randn(x.shape, generator=g)

    def make_step(forward):
        def step():
            h = x
            for p in params:
                h = forward(h, ei, p, flags)
            for p in params:
                for v in p.values():
                    v.grad = None
            h.backward(gout)
        return step

    def median_time(step, budget):
        for _ in range(3):
            t0 = time.perf_counter()
            step()
            one = time.perf_counter() - t0
        k = int(max(3, min(10, budget / max(one, 1e-6))))
        ts = []
        for _ in range(k):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], k

    dt, k = median_time(make_step(oracle.adapter_forward), budget_s * 0.6)
    rdt, rk = median_time(make_step(oracle.adapter_forward_reassociated), budget_s * 0.4)
    pin_host_threads()
    return {"value": round(batch.num_graphs / dt, 1), "unit": "graphs/s", "cores": threads,
            "kind": "port", "ms_per_step": round(dt * 1e3, 3),
            "sample": f"3 warm-up steps, median of {k} steps of the same batch ({batch.num_graphs} graphs, "
                      f"{cfg['adapters']} adapters fwd+bwd, PyG op order, fp32, torch CPU, threads on "
                      f"{threads} physical cores of one NUMA node)",
            "reassociated_value": round(batch.num_graphs / rdt, 1),
            "reassociated_ms_per_step": round(rdt * 1e3, 3),
            "reassociated_sample": f"same, both propagations at rank width (median of {rk})"}


# --------------------------------------------------------------------------- main
def run_workload(name: str, normalize: bool, steps: int, warmup: int, rank: int, world: int,
                 device, with_cpu: bool, strong: bool = False, reps: int = 1) -> dict:
    fresh = ga.synth.WORKLOADS[name]["kind"] == "molecule"   # transductive graphs are static
    if os.environ.get("GCONV_BENCH_STATIC_PLAN"):   # lab knob: one batch, its plan cached (no build beside the step)
        fresh = False
    if RESIDENT and fresh and not strong:
        stream = StoreStream(name, rank, device, RESIDENT)
    else:
        stream = BatchStream(name, rank, device, k=4, fresh=fresh, shard=(rank, world) if strong else None)
    cfg = stream.cfg
    by_rows = PARTITION == "rows" and not fresh and world > 1
    mods = build_adapters(cfg, normalize, device, ga.RowSlabGConvAdapter if by_rows else None)
    if by_rows:   # ONE graph over all ranks: this rank keeps its rows of x / grad_out
        batch, x, ei, gout = stream.items[0]
        slabs = None
        for m in mods:
            slabs = m.partition(x.shape[-2])
        keep = (lambda t: slabs.take(t.reshape(-1, t.shape[-1])).reshape(*t.shape[:-2], slabs.rows, t.shape[-1]))
        stream.items[0] = (batch, keep(x), ei, keep(gout))
    params = list(mods.parameters())
    bucket = ga.dist.FlatGradBucket(params) if world > 1 else None
    mods = list(mods)
    graphs = [0]
    pin_host_threads()   # after the CPU-side generation of the synthetic batches

    def step():
        graphs[0] += adapter_step(mods, params, stream, normalize, bucket)

    # Host-bound batches (the reference's 32-molecule ESOL batches: 0.12 ms of kernels per step, 0.16 of host work): the
    # autograd nodes run on the calling thread — no hand-off to the engine's device thread and back per backward()
    # (tools/stepprof.py esol fresh [st]: backward 83 -> 75 us of host time, step 163 -> 153 us). A training script
    # does the same with one line. GCONV_BENCH_AUTOGRAD_THREADS=1 keeps the engine's threads (A/B).
    host_bound = (not by_rows and isinstance(stream, BatchStream) and not stream.prefetch_beside_backward and world == 1
                  and not os.environ.get("GCONV_BENCH_AUTOGRAD_THREADS"))
    with torch.autograd.set_multithreading_enabled(not host_bound):
        settled = settle(step, device, world)
        graphs[0] = 0
        cfs0 = ga.dist.cpu_throttle_stats()
        dt = timed_steps(step, steps, warmup, world, device)
        for _ in range(reps - 1):   # (extras only: a 15 ms window of a host-bound step is one scheduler hiccup long)
            dt = min(dt, timed_steps(step, steps, 0, world, device))
        cfs1 = ga.dist.cpu_throttle_stats()
    if world > 1:   # what a bad scaling point needs to be read: was the container throttled inside the timed region?
        LAST_DIAG["cfs_throttle_delta_over_timed_region"] = {k: cfs1[k] - cfs0[k] for k in cfs0}
        if bucket is not None:
            bucket.wait()
            LAST_DIAG["all_reduce"] = probe_all_reduce(bucket, world, device)
        pins = [None] * world
        dist.all_gather_object(pins, f"{_PINNED[0]}-{_PINNED[-1]}" if _PINNED else "none")
        LAST_DIAG["pinned_cores_per_rank"] = pins
    ga.PLAN_CACHE.clear()
    if isinstance(stream, StoreStream):
        batch0 = ga.synth.molecule_batch(cfg["mol"], cfg["graphs"], hidden=cfg["d"], seed=rank * 1000)
        edges = round(stream.edges / max(stream.count, 1))
    else:
        batch0 = stream.items[0][0]
        edges = round(sum(b.edge_index.shape[1] for b, *_ in stream.items) / len(stream.items))
    graphs_per_step = stream.avg("num_graphs")
    total_graphs = cfg["graphs"] if strong else graphs_per_step * (1 if by_rows else world)
    res = {"value": total_graphs * steps / dt, "ms_per_step": dt / steps * 1e3,
           "cfg": cfg, "by_rows": by_rows, "strong": strong,
           "nodes_per_rank": round(stream.avg("num_nodes")),
           "edges_per_rank": edges,
           "graphs_per_rank": round(graphs_per_step), "settle_steps": settled}
    if with_cpu:
        res["cpu"] = cpu_baseline(batch0, cfg, normalize)
    return res


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="hiv", choices=sorted(ga.synth.WORKLOADS))
    ap.add_argument("--normalize", type=int, default=1,
                    help="1 = D^-1/2 (A+I) D^-1/2 (north_star subject); 0 = YAML default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--graph-replay", action="store_true",
                    help="(child of the default run) configs[3] adapter step, eager vs one replayed HIP graph: one JSON line")
    ap.add_argument("--banded", action="store_true", help="also time the SpMM on a banded 1M-node graph")
    ap.add_argument("--resident", type=int, default=0, metavar="G",
                    help="keep a G-molecule dataset in HBM and draw a fresh random batch every step "
                         "(device-side collation + plan assembly) instead of rotating 4 fixed batches")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong (default for --workload hiv): the SAME 4096-molecule batches of configs[2] for any N, "
                         "every rank takes 4096 / N graphs (N = 1: the whole batch on one GPU); weak (default for "
                         "the other workloads): every rank runs its own batch (hiv: a 512-molecule shard)")
    ap.add_argument("--partition", default="graphs", choices=["graphs", "rows"],
                    help="N > 1 on a single-graph workload (cora / pubmed / arxiv): 'graphs' runs one replica "
                         "of the graph per GPU, 'rows' splits the ONE graph by rows over the GPUs "
                         "(strong scaling; all-gathers of the rank-width operands between the stages)")
    args = ap.parse_args()
    if args.graph_replay:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (there is no CPU product path)")
        torch.cuda.set_device(0)
        ga._lib.load()
        pin_host_threads()
        res = {}
        for tname in ("cora", "pubmed"):
            res.update(measure_graph_replay(torch.device("cuda", 0), tname))
        print(json.dumps(res))
        return
    global RESIDENT, PARTITION
    RESIDENT = args.resident
    PARTITION = args.partition

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a rank's world comes from torch.distributed.run; `--gpus N` without it is handled by the
        # launcher above (N > 1) — anything else is a mistake in the command line
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; run `python bench.py --gpus N` "
                         f"(it starts the ranks itself) or `python -m torch.distributed.run --nproc-per-node N "
                         f"bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU product path)")
    # one rank per GPU. Fewer GPUs than ranks is an error unless GCONV_BENCH_BACKEND=gloo asks for a
    # functional rehearsal of the N > 1 path (ranks wrap around the GPUs; never a measurement)
    ngpu = torch.cuda.device_count()
    backend = os.environ.get("GCONV_BENCH_BACKEND", "nccl") if world > 1 else "none"
    if world > ngpu and backend != "gloo":
        raise SystemExit(f"bench.py: {world} ranks but {ngpu} GPU(s) visible (GCONV_BENCH_BACKEND=gloo rehearses "
                         f"several ranks per GPU)")
    dev_index = local_rank % max(ngpu, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    devices = [f"cuda:{dev_index} {torch.cuda.get_device_name(dev_index)}"]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        gathered = [None] * world
        dist.all_gather_object(gathered, f"rank {rank}: {devices[0]}")
        devices = gathered
    normalize = bool(args.normalize)
    ga._lib.load()

    solo = world == 1 and rank == 0
    molecule = ga.synth.WORKLOADS[args.workload]["kind"] == "molecule"
    if args.scaling is None:   # hiv: configs[2] is ONE 4096-molecule batch over the GPUs
        args.scaling = "strong" if args.workload in ("hiv", STRONG_WORKLOAD) else "weak"
    strong = args.scaling == "strong" and molecule
    if strong:
        args.workload = STRONG_WORKLOAD
    main_res = run_workload(args.workload, normalize, args.steps, args.warmup, rank, world, device,
                            with_cpu=solo and not args.no_cpu_baseline, strong=strong)
    line = {
        "metric": "adapter fwd+bwd graphs/sec (molecule batches)",
        "value": round(main_res["value"], 1),
        "unit": "graphs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(main_res["ms_per_step"], 4),
        "higher_is_better": True,
        "scaling": "strong" if (main_res["by_rows"] or strong) else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{'M' if main_res['cfg']['kind'] == 'molecule' else 'T'}-{args.workload}: {main_res['graphs_per_rank']} graphs/GPU "
                               f"({main_res['nodes_per_rank']} nodes, {main_res['edges_per_rank']} edges), "
                               f"d={main_res['cfg']['d']}, r={main_res['cfg']['r']}, "
                               f"{main_res['cfg']['adapters']} adapters fwd+bwd, "
                               + ("fresh graph plan per step (built on a side stream two steps ahead, while the previous steps run)"
                                  if main_res["cfg"]["kind"] == "molecule" else "static graph (plan built once)")
                               + (f" + ONE flat grad all-reduce per step over {'RCCL' if backend == 'nccl' else backend} "
                                  "(gradients deposited in the bucket by the backward kernels, collective "
                                  "asynchronous)" if world > 1 else ""),
                   "normalize": normalize,
                   "sharding": (f"ONE graph by rows over {world} rank(s): 4 all-gathers of [N, Fp] per adapter "
                                f"and step + flat grad all-reduce" if main_res["by_rows"]
                                else (f"by graph: the same {main_res['cfg']['graphs']}-molecule batches cut "
                                      f"into {world} shard(s) of {main_res['graphs_per_rank']} (strong scaling)" if strong
                                      else f"by graph, {world} rank(s), one {main_res['graphs_per_rank']}-molecule "
                                           f"shard each (weak scaling; strong-scaling reference at N = 1: "
                                           f"extra.{STRONG_WORKLOAD}_graphs_per_s)")),
                   "backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend),
                   "world_size": dist.get_world_size() if world > 1 else 1,
                   "devices": devices,
                   "settle_steps": main_res["settle_steps"],
                   "host_threads": (f"pinned to cores {_PINNED[0]}-{_PINNED[-1]} (GPU's NUMA node)"
                                    if _PINNED else "not pinned"),
                   # the container's CFS quota in CPUs (None: none) and what one rank pins itself to under it
                   # (gconv_adapter_amd.dist.host_core_budget: floor(quota / ranks), at most 8)
                   "cpu_quota": ga.dist.cpu_quota(),
                   "cores_per_rank": len(_PINNED) if _PINNED else None,
                   "surface": "GConvAdapter nn.Module -> C ABI -> HIP"},
    }
    # what `value` is, in one token, and where the number that rounds 1-3 headlined lives (round-4 advisor finding:
    # the headline moved from the 512-molecule shard to the whole batch in round 4 — values across that change are
    # not like for like)
    line["headline_workload"] = f"{args.workload}_{line['scaling']}_n{world}"
    line["comparable_to_previous_round"] = {
        "round_4": "same metric, workload and scaling (hiv4096 strong): value / ms_per_step compare directly",
        "rounds_1_to_3": "headlined the 512-molecule shard (weak scaling): compare extra.hiv512_graphs_per_s / "
                         "extra.hiv512_ms_per_step"}
    if world > 1 and LAST_DIAG:
        per = LAST_DIAG.get("ms_per_step_per_rank") or [main_res["ms_per_step"]]
        line["diagnostics"] = {"ms_per_step_min_over_ranks": min(per), "ms_per_step_max_over_ranks": max(per), **LAST_DIAG,
                               "note": "per-rank times are each rank's own K steps before the closing barrier; the all-reduce "
                                       "probe runs after the timed region; cfs = cpu.stat deltas of the container over the timed region"}
    if solo:
        extra = {}
        if "cpu" in main_res:
            line["cpu_baseline"] = main_res["cpu"]
            extra["speedup_vs_cpu"] = round(main_res["value"] / main_res["cpu"]["value"], 1)
            #  the same ratio against the CPU running the REASSOCIATED order: without the
            # reference's d-wide second propagation in the denominator
            extra["speedup_vs_cpu_reassociated"] = round(main_res["value"] / main_res["cpu"]["reassociated_value"], 1)
        if not args.no_extra:
            # the YAML default of the reference (configs/finetune/gconv_adapter.yaml:8, normalize: false)
            # / the north-star subject (normalize: true): whichever the headline did not run
            o = run_workload(args.workload, not normalize, max(20, args.steps // 2), args.warmup, rank, world,
                             device, with_cpu=False, strong=strong)
            key = "normalize_true" if not normalize else "normalize_false_yaml_default"
            extra[f"{key}_graphs_per_s"] = round(o["value"], 1)
            extra[f"{key}_ms_per_step"] = round(o["ms_per_step"], 4)
            if not strong and molecule:
                # strong-scaling reference: the whole 4096-molecule batch of configs[2] on this ONE GPU
                # (fresh plan per step, through the module). value(N = 8, weak) / this = speed-up at 8 GPUs
                o = run_workload(STRONG_WORKLOAD, normalize, max(20, args.steps // 4), args.warmup, rank, world,
                                 device, with_cpu=False, strong=True)
                extra[f"{STRONG_WORKLOAD}_graphs_per_s"] = round(o["value"], 1)
                extra[f"{STRONG_WORKLOAD}_ms_per_step"] = round(o["ms_per_step"], 4)
                pin_host_threads()
            if strong:
                # the per-GPU unit at N = 8 (one 512-molecule shard; the round 1-3 headline) on this ONE GPU:
                # ms_per_step here + one ~100 KB all-reduce is what a rank of the 8-GPU run has to do
                o = run_workload("hiv", normalize, args.steps, args.warmup, rank, world, device, with_cpu=False, reps=3)
                extra["hiv512_graphs_per_s"] = round(o["value"], 1)
                extra["hiv512_ms_per_step"] = round(o["ms_per_step"], 4)
                extra["hiv512_projected_speedup_at_8_gpus_without_collective"] = round(
                    main_res["ms_per_step"] / o["ms_per_step"], 2)
                pin_host_threads()
            ax = measure_arxiv(device, banded_control=args.banded)
            line["roofline"] = ax["roofline"]
            extra.update(ax["extra"])
            extra.update(measure_tile_kernels(device))
            extra.update(measure_gin(device))
            extra.update(measure_arxiv_projections(device))
            extra.update(measure_graph_build(device))
            extra.update(measure_no_prefetch_step(device, normalize))
            for tname in ("cora", "pubmed"):
                extra.update(measure_transductive(device, tname))
            extra.update(graph_replay_child())
            other = "esol" if args.workload != "esol" else "hiv"
            o = run_workload(other, normalize, max(20, min(args.steps, 200) // 2), args.warmup, rank, world,
                             device, with_cpu=not args.no_cpu_baseline, reps=3)
            extra[f"{other}_graphs_per_s"] = round(o["value"], 1)
            extra[f"{other}_ms_per_step"] = round(o["ms_per_step"], 4)
            if "cpu" in o:
                extra[f"{other}_cpu_graphs_per_s"] = o["cpu"]["value"]
                extra[f"{other}_speedup_vs_cpu"] = round(o["value"] / o["cpu"]["value"], 1)
            pin_host_threads()
            ga.PLAN_CACHE.clear()
            torch.cuda.empty_cache()   # the S-arxiv phases leave ~1 GB of cached blocks of odd sizes behind
            extra.update(measure_finetune_step(device, normalize))   # last: torch layers, allocator churn
            extra.update(measure_finetune_step(device, normalize, name="hiv4096", steps=30))   # the whole batch
            extra["best_of_3_averages"] = ["arxiv_proj_*", "cora_*", "pubmed_*", "*_graph_build_wall_us",
                                           "*_no_prefetch_*", "*_finetune_step_*", "arxiv_step4_*",
                                           f"{other}_ms_per_step", "hiv512_ms_per_step"]   # launch-bound extras: see best_event_time_ms
        line["extra"] = extra
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
