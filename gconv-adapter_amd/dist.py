"""Graph-sharded data parallelism for the adapter path (one process per GPU).

The reference is single-process / single-device (``torch.device("cuda:0")``,
``/root/reference/scripts/finetune_inductive_learning.py:179``); this module is
new. Molecule batches are block-diagonal, so a batch is partitioned BY GRAPH
across ranks with no halo and no data-path collective
(``synth.shard_graphs``); the only exchange per optimizer step is one
all-reduce of the trainable-parameter gradients.

That set is small — adapters + head + norm affine params, ~29 k fp32 ≈ 116 KB
for HIV / rank 8 (freeze policy ``src/models/inductive/gnn.py:230-241``) — so
the all-reduce is latency-bound on xGMI: ONE flat bucket, ONE RCCL call per
step. ``FlatGradBucket`` owns one contiguous buffer; by the time its collective
is launched every ``p.grad`` is a view into it — written there directly by the
adapters' backward (gradient sinks), accumulated in place by autograd
(``bind()``), or packed with one multi-tensor copy (parameters autograd fills
with fresh tensors: head, norm layers) — and nothing is copied back.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: Optional[str] = None) -> int:
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Backend "nccl" is RCCL on ROCm; "gloo" is used for the CPU logic tests."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", "0")),
                            world_size=world)
    return world


def adapter_trainable_parameters(model: torch.nn.Module) -> List[torch.nn.Parameter]:
    """Trainable set under the reference's freeze rule: a parameter trains iff its
    name contains 'adapter' or it belongs to the prediction head / a norm layer
    (``gnn.py:230-241``; ``nodeformer.py:377-388``). Here: every parameter that
    currently requires grad, in ``named_parameters`` order (identical on all ranks)."""
    return [p for _, p in model.named_parameters() if p.requires_grad]


class FlatGradBucket:
    """One contiguous gradient buffer for a parameter list; every parameter starts on a 16-byte
    boundary (padding floats are zero and stay zero).

    Three ways to get the gradients into it, cheapest first:

    * ``attach_sinks()`` (the default on a HIP device with the C++ binding): every view is registered
      as the parameter's *gradient sink*. The adapter's backward node then lets its last kernel (the
      fixed-order reduction of the per-workgroup partials) write ``dW / db / ds`` STRAIGHT into the
      bucket and sets ``p.grad`` to the view — no ``cat`` before the collective, no copy back after
      it. Needs the step to start from ``p.grad = None`` (``zero_grad(set_to_none=True)``); a
      parameter that still holds a gradient is accumulated into the ordinary way.
    * parameters that autograd fills itself (prediction head, norm layers: fresh tensors per step)
      are packed by ``all_reduce_grads`` with ONE multi-tensor copy and re-bound to their views.
    * ``bind()``: every ``p.grad`` a view up front, autograd accumulates in place (``zero_()`` per step).

    ``all_reduce_grads(async_op=True)`` launches the ONE collective of the step behind the current
    stream's work and returns; ``wait()`` — before the optimizer, or before the next backward writes
    the sinks again — makes the current stream wait for it. RCCL runs it on its own stream, so it is
    off the critical path of whatever the host queues in between (the next batch's graph build)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], sinks: bool = True):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatGradBucket needs at least one parameter")
        dev, dt = self.params[0].device, self.params[0].dtype
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("all bucketed parameters must share device and dtype")
        self.offsets = []
        total = 0
        for p in self.params:
            self.offsets.append(total)
            total += -(-p.numel() // 4) * 4
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.views = [self.flat[off:off + p.numel()].view_as(p)
                      for p, off in zip(self.params, self.offsets)]
        self._work = None
        self._pending_div = 0
        self.sinks = False
        # sinks need the C++ binding; without it (ctypes path, CPU tensors) every p.grad is bound up front
        if not (sinks and dev.type == "cuda" and dt == torch.float32 and self.attach_sinks()):
            self.bind()

    def attach_sinks(self) -> bool:
        """Registers the views as gradient sinks with the C++ binding (see the class docstring).
        False when the binding is absent (ctypes path): use ``bind()`` or plain ``all_reduce_grads``."""
        from . import _torch_ext
        ext = _torch_ext.get()
        if ext is None or not hasattr(ext, "set_grad_sink"):
            return False
        for p, v in zip(self.params, self.views):
            ext.set_grad_sink(p, v)
        self.sinks = True
        return True

    def detach_sinks(self) -> None:
        from . import _torch_ext
        ext = _torch_ext.get()
        if self.sinks and ext is not None:
            for p, v in zip(self.params, self.views):
                ext.set_grad_sink(p, None, v)      # (only if the registered sink is still this bucket's)
        self.sinks = False

    def __del__(self):
        try:
            self.detach_sinks()
        except Exception:   # interpreter shutdown
            pass

    def bind(self) -> None:
        """Makes every ``p.grad`` a view into the flat buffer (autograd then accumulates in
        place: one small add kernel per parameter and backward)."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero_(self) -> None:
        """Replaces ``optimizer.zero_grad()`` in ``bind()`` mode: keeps the views, zeroes in one op."""
        self.flat.zero_()

    def wait(self) -> None:
        """Orders the current stream behind the collective launched by an ``async_op`` call (and
        applies the division of an ``average=True`` call, which has to follow the sum)."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._pending_div:
            self.flat.div_(self._pending_div)
            self._pending_div = 0

    def all_reduce(self, average: bool = False, async_op: bool = False):
        """Sum (or mean) of the gradients over all ranks: one collective per step. Gradients that
        are not in the bucket yet (a parameter without a sink, or one autograd filled with its own
        tensor) are packed first (``gather_stragglers``): whatever mode the bucket is in, what is
        reduced is every bucketed parameter's gradient, and afterwards every ``p.grad`` is a view.

        With the reference's loss normalisation (sum over valid targets divided by
        the batch's count, ``finetune_inductive_learning.py:62-63``) each rank should
        divide by the GLOBAL count (see ``global_count``) and use ``average=False``.
        ``average=True`` with ``async_op=True`` divides in ``wait()``."""
        self.wait()           # (a collective still in flight from the previous step)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None       # nothing to exchange: gradients stay where autograd / the sinks put them
        self.gather_stragglers()
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            self._work = work
            self._pending_div = dist.get_world_size() if average else 0
        elif average:
            self.flat.div_(dist.get_world_size())
        return work

    def gather_stragglers(self) -> int:
        """Gradients that are not in the bucket yet (autograd's own fresh tensors: parameters without
        a sink, or a step that accumulated) are copied in with ONE multi-tensor copy and ``p.grad`` is
        re-bound to the view. Returns how many there were (0 for an adapter-only parameter list with
        sinks: nothing is launched)."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                raise RuntimeError("FlatGradBucket: every bucketed parameter needs a gradient")
            if g.data_ptr() != v.data_ptr():
                src.append(g)
                dst.append(v)
                p.grad = v
        if src:
            torch._foreach_copy_(dst, src)
        return len(src)

    def all_reduce_grads(self, average: bool = False, async_op: bool = False):
        """The step's gradient exchange for loops that start from ``p.grad = None``: stragglers in
        (see ``gather_stragglers``), ONE collective on the flat buffer, ``p.grad`` = views of it —
        nothing is copied back. ``async_op``: returns at once, ``wait()`` later."""
        return self.all_reduce(average=average, async_op=async_op)


# ---------------------------------------------------------------------------------------
# Host-thread placement. The adapter step on molecule batches is launch-bound: three host
# threads (training loop, autograd engine, graph-plan worker) hand work to each other every
# few microseconds. Left free to roam a 2 x 64-core host they are migrated across CCDs and
# NUMA nodes and the same step runs 1.4–1.7x slower (ESOL: 265 vs 157 us) until the scheduler
# happens to settle. One process per GPU, pinned to a few cores of the GPU's NUMA node, is the
# stable configuration — what `numactl --cpunodebind` does for the reference's single process.
# ---------------------------------------------------------------------------------------
def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _gpu_numa_node(index: int) -> int:
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as fh:
            return max(int(fh.read().strip()), 0)
    except Exception:   # no sysfs / no such attribute: treat the host as one node
        return 0


def set_process_affinity(cpus) -> None:
    """Applies a CPU set to EVERY thread of this process (the engine / worker / runtime threads
    that already exist keep their own masks otherwise)."""
    cpus = set(cpus)
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), cpus)
        except OSError:
            pass


def local_core_set(local_rank: int = 0, local_world: int = 1, cores_per_rank: int = 8) -> List[int]:
    """`cores_per_rank` physical cores of the NUMA node of GPU `local_rank`, disjoint from the
    sets of the other local ranks on that node (each rank computes the same partition)."""
    allowed = sorted(os.sched_getaffinity(0))
    nodes = [_gpu_numa_node(g) for g in range(max(local_world, local_rank + 1))] \
        if torch.cuda.is_available() else [0] * max(local_world, local_rank + 1)
    node = nodes[local_rank]
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as fh:
            node_cpus = [c for c in _parse_cpulist(fh.read()) if c in set(allowed)]
    except OSError:
        node_cpus = allowed
    if not node_cpus:
        node_cpus = allowed
    # physical cores first: the second half of a node's list are the SMT siblings
    phys = node_cpus[:max(len(node_cpus) // 2, 1)] if len(node_cpus) >= 2 * cores_per_rank else node_cpus
    slot = sum(1 for g in range(local_rank) if nodes[g] == node)
    per = max(1, min(cores_per_rank, len(phys) // max(1, nodes.count(node))))
    if local_world == 1:
        # A single process per container: every tenant of a shared host that reasons like this
        # function would land on the SAME first cores of the node (seen as 0.23 -> 0.29 ms steps when
        # two boxes of one host ran at once). Take the least busy aligned block of cores instead.
        return _least_busy_block(phys, per)
    start = (slot * per) % max(len(phys) - per + 1, 1)
    return phys[start:start + per]


def _cpu_busy_jiffies() -> dict:
    busy = {}
    try:
        with open("/proc/stat") as fh:
            for line in fh:
                if line.startswith("cpu") and line[3].isdigit():
                    f = line.split()
                    vals = [int(v) for v in f[1:9]]
                    busy[int(f[0][3:])] = sum(vals) - vals[3] - vals[4]   # everything but idle and iowait
    except (OSError, ValueError, IndexError):
        pass
    return busy


def _least_busy_block(cpus: List[int], per: int, sample_s: float = 0.05) -> List[int]:
    """The aligned block of `per` consecutive entries of `cpus` (one CCD's worth of cores share an L3)
    that was least busy over a short sample of /proc/stat; the first block when nothing can be read."""
    import time
    blocks = [cpus[i:i + per] for i in range(0, len(cpus) - per + 1, per)] or [cpus[:per]]
    a = _cpu_busy_jiffies()
    if not a:
        return blocks[0]
    time.sleep(sample_s)
    b = _cpu_busy_jiffies()
    load = [sum(b.get(c, 0) - a.get(c, 0) for c in blk) for blk in blocks]
    return blocks[min(range(len(blocks)), key=lambda i: (load[i], i))]


# ---------------------------------------------------------------------------------------
# The container's CPU share (cgroup). A GPU box hands a container a CFS QUOTA (the 1-GPU boxes of this
# pool: 16 CPUs of a 256-thread host): a process group that uses more CPU time than quota x period is
# frozen — every thread of it — until the next 100 ms period. One rank of the adapter step keeps three
# host threads busy (training loop, autograd engine, graph-plan worker; RCCL adds a proxy thread), so
# 8 ranks x 3 threads on a 16-CPU quota is 1.5x the share: the ranks' threads must fit INSIDE the quota
# by construction. `host_core_budget` is that rule, `pin_to_local_cores` applies it, and the bench line
# of an N > 1 run carries the quota and the throttle counters so that a bad curve can be told from a bad box.
# ---------------------------------------------------------------------------------------
def parse_cpu_max(text: str) -> Optional[float]:
    """CPUs of a cgroup-v2 ``cpu.max`` line (``"1600000 100000"`` -> 16.0); ``"max 100000"`` (no quota),
    empty or malformed text -> None."""
    f = text.split()
    if len(f) != 2 or f[0] == "max":
        return None
    try:
        quota, period = int(f[0]), int(f[1])
    except ValueError:
        return None
    return quota / period if quota > 0 and period > 0 else None


def parse_cpu_stat(text: str) -> dict:
    """``cpu.stat`` of cgroup v2 (``nr_periods``, ``nr_throttled``, ``throttled_usec``) or v1
    (``throttled_time`` in ns) -> {nr_periods, nr_throttled, throttled_usec}; missing keys are 0."""
    kv = {}
    for line in text.splitlines():
        f = line.split()
        if len(f) == 2:
            try:
                kv[f[0]] = int(f[1])
            except ValueError:
                pass
    usec = kv.get("throttled_usec", kv.get("throttled_time", 0) // 1000)
    return {"nr_periods": kv.get("nr_periods", 0), "nr_throttled": kv.get("nr_throttled", 0), "throttled_usec": usec}


def cpu_quota() -> Optional[float]:
    """This container's CPU quota in CPUs (cgroup v2 ``/sys/fs/cgroup/cpu.max``, v1 ``cpu.cfs_quota_us`` /
    ``cpu.cfs_period_us``), None when there is none or it cannot be read."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            return parse_cpu_max(fh.read())
    except OSError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as q, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as p:
            return parse_cpu_max(f"{q.read().strip()} {p.read().strip()}".replace("-1 ", "max "))
    except OSError:
        return None


def cpu_throttle_stats() -> dict:
    """Throttle counters of this container's cgroup (all zero when unreadable): sample before and after a timed
    region — a growing ``nr_throttled`` means the ranks' host threads were frozen by the CFS quota inside it."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            with open(path) as fh:
                return parse_cpu_stat(fh.read())
        except OSError:
            continue
    return parse_cpu_stat("")


def host_core_budget(local_world: int, quota: Optional[float] = None, most: int = 8) -> int:
    """Cores one rank pins itself to: `most` (8: room for its three busy threads, the runtime's and RCCL's helpers
    on one CCD) unless the container's quota is smaller than `most` x ranks — then floor(quota / ranks), at least
    1: 8 ranks under a 16-CPU quota get 2 cores each, so that all ranks together cannot exceed the quota and no
    CFS period ends in a freeze. Threads then share cores (hand-offs cost a context switch, a few microseconds)
    instead of the whole process group losing the rest of a 100 ms period."""
    if quota is None:
        quota = cpu_quota()
    if quota is None:
        return most
    return max(1, min(most, int(quota // max(local_world, 1))))


def pin_to_local_cores(local_rank: Optional[int] = None, local_world: Optional[int] = None,
                       cores_per_rank: Optional[int] = None) -> List[int]:
    """Pins all threads of this process to `local_core_set(...)`; returns the set. Ranks and
    world default to LOCAL_RANK / LOCAL_WORLD_SIZE (torchrun), `cores_per_rank` to `host_core_budget`
    (GCONV_CORES_PER_RANK overrides). No-op (returns []) when GCONV_NO_PIN is set or the platform has
    no sched_setaffinity."""
    if os.environ.get("GCONV_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return []
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if cores_per_rank is None:
        cores_per_rank = int(os.environ.get("GCONV_CORES_PER_RANK", 0)) or host_core_budget(local_world)
    cpus = local_core_set(local_rank, local_world, cores_per_rank)
    if cpus:
        set_process_affinity(cpus)
    return cpus


def global_count(local_count: torch.Tensor) -> torch.Tensor:
    """All-reduced number of valid targets, so that per-shard losses are divided by
    the count of the whole batch as the single-process reference does."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        local_count = local_count.clone()
        dist.all_reduce(local_count, op=dist.ReduceOp.SUM)
    return local_count


def shard_bounds(num_graphs: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of graphs owned by ``rank``."""
    return (num_graphs * rank) // world, (num_graphs * (rank + 1)) // world
