"""CPU tests of the host side (`-m "not gpu"`): module surface mirrors the reference,
Hydra-style construction, the C-ABI library loads and exports every declared
symbol, and the product path refuses to run without a HIP device."""
from __future__ import annotations

import ctypes
import os
import re

import pytest
import torch

import gconv_adapter_amd as ga

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- constructor / state_dict surface (reference gconv_adapter.py:23-78) ----------------
def test_default_kwargs_match_reference():
    m = ga.GConvAdapter(hidden_size=300, bottleneck_size=16)
    assert m.normalize is True and m.skip_connection is True
    assert m.scalar is None and m.normalization is None
    assert isinstance(m.act_fn, torch.nn.ReLU)


def test_state_dict_keys_and_shapes():
    m = ga.GConvAdapter(300, 8, learnable_scalar=True)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        "conv_down.lin.weight": (8, 300), "conv_down.bias": (8,),
        "conv_up.lin.weight": (300, 8), "conv_up.bias": (300,), "scalar": (1,)}
    assert "scalar" not in ga.GConvAdapter(300, 8).state_dict()
    ln = ga.GConvAdapter(32, 4, normalization="layer_norm").state_dict()
    assert "normalization.weight" in ln and "normalization.bias" in ln
    bn = ga.GConvAdapter(32, 4, normalization="batch_norm").state_dict()
    assert "normalization.running_mean" in bn
    # parameter count 2rd + r + d + 1
    assert sum(p.numel() for p in m.parameters()) == 2 * 8 * 300 + 8 + 300 + 1
    assert all(isinstance(p, torch.nn.Parameter) for p in m.parameters())


def test_near_identity_init():
    torch.manual_seed(0)
    m = ga.GConvAdapter(300, 16, learnable_scalar=True)
    assert m.conv_down.lin.weight.abs().max() < 1e-4 and m.conv_down.lin.weight.std() > 1e-6
    assert m.conv_up.lin.weight.abs().max() < 1e-4
    assert torch.count_nonzero(m.conv_down.bias) == 0 and torch.count_nonzero(m.conv_up.bias) == 0
    assert m.scalar.item() == 1.0


def test_invalid_enums_raise_value_error_with_reference_messages():
    with pytest.raises(ValueError, match="Invalid conv_type"):
        ga.GConvAdapter(8, 2, conv_type="gin")
    with pytest.raises(ValueError, match="Invalid non_linearity"):
        ga.GConvAdapter(8, 2, non_linearity="gelu")
    with pytest.raises(ValueError, match="Invalid normalization"):
        ga.GConvAdapter(8, 2, normalization="group_norm")
    for dead in ("sage", "gat"):        # dead code in the reference
        with pytest.raises(NotImplementedError):
            ga.GConvAdapter(8, 2, conv_type=dead)


def test_load_state_dict_roundtrip_and_freeze_rule():
    a = ga.GConvAdapter(32, 4, learnable_scalar=True)
    b = ga.GConvAdapter(32, 4, learnable_scalar=True)
    with torch.no_grad():
        for p in a.parameters():
            p.normal_()
    b.load_state_dict(a.state_dict(), strict=True)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    # the reference freezes by parameter NAME containing 'adapter' (gnn.py:230-241)
    host = torch.nn.Module()
    host.pre_adapters = torch.nn.ModuleDict({"0": a})
    host.backbone = torch.nn.Linear(4, 4)
    for name, p in host.named_parameters():
        p.requires_grad = "adapter" in name
    trainable = ga.dist.adapter_trainable_parameters(host)
    assert len(trainable) == 5 and all(p.requires_grad for p in trainable)


# ---- Hydra-style construction (gnn.py:79,82; configs/finetune/gconv_adapter.yaml) -------
def test_yaml_schema_matches_reference_and_instantiates():
    cfg = ga.load_finetune_config("gconv_adapter")
    assert set(cfg) == {"adapter", "positions", "type"}
    assert set(cfg["adapter"]) == {"_target_", "bottleneck_size", "conv_type", "learnable_scalar",
                                   "non_linearity", "normalization", "normalize", "skip_connection"}
    assert cfg["adapter"]["_target_"] == "gconv_adapter_amd.GConvAdapter"
    assert cfg["positions"] == ["pre"] and cfg["type"] == "sequential"
    m = ga.instantiate(cfg["adapter"], hidden_size=300)      # call-site kwarg, as add_adapter does
    assert isinstance(m, ga.GConvAdapter)
    assert m.hidden_size == 300 and m.bottleneck_size == 16
    assert m.normalize is False and m.scalar is not None     # YAML overrides the ctor defaults


def test_insertion_contract_moduledict_per_layer():
    cfg = ga.load_finetune_config()
    pre = torch.nn.ModuleDict()
    for layer in range(5):
        pre[str(layer)] = ga.instantiate(cfg["adapter"], hidden_size=300)
    names = [n for n, _ in pre.named_parameters()]
    assert "0.conv_down.lin.weight" in names and "4.scalar" in names


# ---- no CPU product path ----------------------------------------------------------------
def test_forward_on_cpu_raises_runtime_error():
    m = ga.GConvAdapter(8, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.randn(3, 8), torch.tensor([[0, 1], [1, 2]]))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ga.GraphPlan(torch.tensor([[0, 1], [1, 2]]), 3, True)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gconv-adapter_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "oracle/" not in text or f.endswith(".md"), f


# ---- C ABI: library loads, every declared symbol is exported ------------------------------
def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gconv_adapter.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gconv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    declared = _declared_functions()
    assert len(declared) >= 13
    lib = ctypes.CDLL(ga._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/gconv_adapter.h but not exported"
    assert sorted(ga._lib.SIGNATURES) == declared, "ctypes binding out of sync with the header"


def test_library_host_only_entry_points():
    lib = ga._lib.load()
    assert lib.gconv_abi_version() == 11
    assert [lib.gconv_padded_rank(r) for r in (0, 1, 4, 5, 8, 16, 17, 33, 64, 65)] == \
        [0, 4, 4, 8, 8, 16, 32, 64, 64, 0]
    assert b"workspace" in lib.gconv_status_string(-3)
    assert lib.gconv_padded_rank(64) == 64
    with pytest.raises(RuntimeError, match="invalid argument"):
        ga._lib.check(-1, "x")


def test_header_constants_match_binding():
    text = open(os.path.join(ROOT, "include", "gconv_adapter.h")).read()
    assert int(re.search(r"#define GCONV_LONG_ROW\s+(\d+)", text).group(1)) == ga._lib.LONG_ROW
    assert int(re.search(r"#define GCONV_SEG_EDGES\s+(\d+)", text).group(1)) == ga._lib.SEG_EDGES
    for name in ("TILE_NODES", "TILE_NODES_MAX", "TILE_HALO_NODES", "TILE_HALO_EDGES", "TILE_HALO_SLOT"):
        assert int(re.search(r"#define GCONV_%s\s+(\d+)" % name, text).group(1)) == getattr(ga._lib, name), name


def test_nominal_tile_size_rule():
    """`gconv_tile_nodes` (the library) == `_lib.tile_nodes` (the binding's restatement): about
    N / 256 nodes, one tile per persistent workgroup, k equal tiles each beyond 256 x 64 nodes."""
    lib = ga._lib.load()
    for n in (0, 1, 31, 431, 8191, 8193, 13104, 16384, 16385, 20000, 102952, 170000, 1 << 20):
        assert lib.gconv_tile_nodes(n) == ga._lib.tile_nodes(n), n
    assert ga._lib.tile_nodes(431) == 16 and ga._lib.tile_nodes(13104) == 52
    assert ga._lib.tile_nodes(16385) == 65 and ga._lib.tile_nodes(102952) == 101
    assert ga._lib.tile_slack(52, 13104) == 13 and ga._lib.tile_slack(101, 102952) == 64
    assert ga._lib.tile_slack(16, 431) == 4


# ---- synthetic workloads ---------------------------------------------------------------
def test_molecule_batches_are_block_diagonal_and_symmetric():
    b = ga.synth.molecule_batch("hiv", 64, seed=3)
    ei = b.edge_index
    sizes = b.graph_ptr[1:] - b.graph_ptr[:-1]
    gid = torch.repeat_interleave(torch.arange(64), sizes)
    assert (gid[ei[0]] == gid[ei[1]]).all() and (ei[0] != ei[1]).all()
    fwd = set(zip(ei[0].tolist(), ei[1].tolist()))
    assert all((t, s) in fwd for s, t in fwd)
    assert sizes.min() >= 2 and sizes.max() <= 222


def test_shard_graphs_partitions_the_batch():
    b = ga.synth.molecule_batch("esol", 32, seed=1)
    shards = [ga.synth.shard_graphs(b, r, 4) for r in range(4)]
    assert sum(s.num_graphs for s in shards) == 32
    assert sum(s.num_nodes for s in shards) == b.num_nodes
    assert sum(s.edge_index.shape[1] for s in shards) == b.edge_index.shape[1]
    assert torch.equal(torch.cat([s.x for s in shards]), b.x)
    for s in shards:
        assert s.edge_index.min() >= 0 and s.edge_index.max() < s.num_nodes


def test_workload_shapes():
    b, cfg = ga.synth.make_workload("esol")
    assert b.num_graphs == 32 and 300 < b.num_nodes < 600 and cfg["r"] == 4
    b, cfg = ga.synth.make_workload("cora")
    assert b.x.shape == (1, 2708, 32) and cfg["r"] == 16


def test_cpp_autograd_binding_loads_and_binds_the_same_c_abi():
    ext = ga._torch_ext.get()
    assert ext is not None, "lib/_gconv_torch.so missing: run __graft_entry__.build()"
    assert hasattr(ext, "adapter_apply") and hasattr(ext, "load_library")
    # CPU tensors are refused by the binding as well (RuntimeError, no compute fallback)
    with pytest.raises(RuntimeError):
        ext.adapter_apply(torch.randn(2, 8), torch.randn(2, 8), None, torch.randn(8, 2), None, None,
                          torch.zeros(1, dtype=torch.uint8), 0, 4, 16, 16, 1, True, 0, 0)


def test_host_thread_pinning_helpers():
    """dist.local_core_set: every local rank gets a non-empty set inside the process's CPU mask,
    ranks on the same NUMA node get disjoint sets when there are enough cores."""
    import os
    from gconv_adapter_amd import dist as gd
    assert gd._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = set(os.sched_getaffinity(0))
    sets = [gd.local_core_set(r, 4, cores_per_rank=2) for r in range(4)]
    for s in sets:
        assert s and set(s) <= allowed
    if len(allowed) >= 16:
        flat = [c for s in sets for c in s]
        assert len(flat) == len(set(flat))
    os.environ["GCONV_NO_PIN"] = "1"
    try:
        assert gd.pin_to_local_cores() == []
    finally:
        del os.environ["GCONV_NO_PIN"]


def test_ctypes_structs_match_the_header():
    """Field names and order of the POD structs in include/gconv_adapter.h == the ctypes mirrors
    (a silent drift here would hand the kernels shifted pointers)."""
    text = open(os.path.join(ROOT, "include", "gconv_adapter.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)

    def fields(name):
        chunk = [c for c in text.split("typedef struct") if re.search(r"\}\s*%s;" % name, c)][0]
        body = chunk[chunk.index("{") + 1:chunk.index("}")]
        return [m.group(1) for m in re.finditer(r"\b(\w+)(?:\[\d+\])?\s*;", body)]

    for cname, ctype in (("gconv_csr_t", ga._lib.CsrStruct), ("gconv_graph_t", ga._lib.GraphStruct),
                         ("gconv_tiling_t", ga._lib.TilingStruct),
                         ("gconv_adapter_params_t", ga._lib.AdapterParams),
                         ("gconv_pending_finalize_t", ga._lib.PendingFinalize)):
        assert fields(cname) == [f[0] for f in ctype._fields_], cname
    # and the sizes the C compiler sees (from the library, which was built from the same header)
    assert ctypes.sizeof(ga._lib.PendingFinalize) == 24 * 8
    lib = ga._lib.load()
    for i, ctype in enumerate((ga._lib.CsrStruct, ga._lib.GraphStruct, ga._lib.AdapterParams,
                               ga._lib.PendingFinalize)):
        assert lib.gconv_abi_struct_bytes(i) == ctypes.sizeof(ctype)
    assert lib.gconv_abi_struct_bytes(4) == 32 == 4 * len(ga.GraphPlan.TILE_FIELDS)
    assert lib.gconv_abi_struct_bytes(5) == ctypes.sizeof(ga._lib.TilingStruct)


def test_batch_norm_module_is_a_drop_in_on_any_input():
    """`gconv_adapter_amd.BatchNorm1d` is an nn.BatchNorm1d: same state dict, same results wherever the
    HIP kernels do not apply (CPU tensors here), and the freeze rule's isinstance test sees it."""
    import torch
    import gconv_adapter_amd as ga
    torch.manual_seed(0)
    ref, mine = torch.nn.BatchNorm1d(12), ga.BatchNorm1d(12)
    relu = ga.BatchNorm1d(12, fused_relu=True)
    assert isinstance(mine, torch.nn.BatchNorm1d) and list(mine.state_dict()) == list(ref.state_dict())
    mine.load_state_dict(ref.state_dict()); relu.load_state_dict(ref.state_dict())
    x = torch.randn(9, 12)
    want = ref(x)
    assert torch.equal(mine(x), want) and torch.equal(relu(x), torch.relu(want))
    assert torch.equal(mine.running_mean, ref.running_mean) and int(mine.num_batches_tracked) == 1
    adapter = ga.GConvAdapter(8, 2, normalization="batch_norm")
    assert isinstance(adapter.normalization, torch.nn.BatchNorm1d)
    assert set(k for k in adapter.state_dict() if k.startswith("normalization")) == {
        "normalization.weight", "normalization.bias", "normalization.running_mean", "normalization.running_var",
        "normalization.num_batches_tracked"}


def test_copies_drop_the_per_instance_caches():
    """deepcopy / pickle of a module that has run: the ParamPack handle, the cached Parameter
    tuple and the anchor tensor belong to the ORIGINAL and must not travel with the copy
    (the copy would release the original's pack on its first forward)."""
    import copy
    import pickle
    mod = ga.GConvAdapter(32, 4, learnable_scalar=True)
    mod.__dict__["_flat_params"] = ("stale",)
    mod.__dict__["_anchor"] = torch.zeros(1)
    for twin in (copy.deepcopy(mod), pickle.loads(pickle.dumps(mod))):
        assert not any(k in twin.__dict__ for k in ga.GConvAdapter._TRANSIENT)
        assert set(twin.state_dict()) == set(mod.state_dict())
        assert twin.conv_down.lin.weight is not mod.conv_down.lin.weight
    assert "_flat_params" in mod.__dict__
    del mod.__dict__["_flat_params"], mod.__dict__["_anchor"]


def test_row_slab_adapter_copies_drop_the_process_group():
    """A RowSlabGConvAdapter partitioned over the default group keeps its slabs through a deepcopy;
    one partitioned over an explicit group drops handle and slabs (the copy must call partition())."""
    import copy
    m = ga.RowSlabGConvAdapter(32, 4)
    m.partition(100, world=2, rank=1)
    c = copy.deepcopy(m)
    assert (c.__dict__["_slabs"].lo, c.__dict__["_slabs"].hi) == (50, 100) and "_group" not in c.__dict__
    m.__dict__["_group"] = object()           # stands in for a ProcessGroup (not picklable)
    c = copy.deepcopy(m)
    assert "_slabs" not in c.__dict__ and "_group" not in c.__dict__


# ---- bench.py as its own launcher (`python bench.py --gpus N`, no torchrun around it) --------------
def _bench_dryrun(args, env_extra=None):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(GCONV_BENCH_LAUNCH_DRYRUN="1", **(env_extra or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_gpus_n_starts_n_child_ranks_without_loading_torch():
    """`--gpus N` (N > 1, no WORLD_SIZE) = launcher: the N ranks are CHILDREN started through
    torch.distributed.run on 127.0.0.1, and the launcher has loaded neither torch nor the package
    (nothing in it can have initialised HIP) when it starts them."""
    import sys
    for flag in (["--gpus", "8"], ["--gpus=8"]):
        got = _bench_dryrun(flag + ["--steps", "7", "--warmup", "3"])
        cmd = got["launcher_cmd"]
        assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
        assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
        script = cmd.index(os.path.join(ROOT, "bench.py"))
        assert cmd[script + 1:] == flag + ["--steps", "7", "--warmup", "3"]      # the ranks get the same flags
        assert got["torch_loaded"] is False and got["package_loaded"] is False
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("\nimport torch")]
    assert "sys.exit(launch_ranks(" in head, "the launcher must run before torch is imported"
    assert "os.exec" not in src and "execv" not in src


def test_bench_rank_role_is_not_a_launcher():
    """Under torch.distributed.run (WORLD_SIZE set) the same command line is a rank, and --gpus that
    contradicts WORLD_SIZE is refused instead of silently measuring another world."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", GCONV_BENCH_LAUNCH_DRYRUN="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "launcher_cmd" not in out.stdout
    assert "WORLD_SIZE=2" in out.stderr


def test_header_is_plain_c_and_a_c_program_links_the_library():
    """`include/gconv_adapter.h` compiles as C11 (-Wall -Werror) and `tools/cabi_spmm_bench.c` — graph build +
    SpMM from plain C, no torch, no C++ — links against the library (a C ABI usable without Python)."""
    import __graft_entry__ as ge
    out = ge.build_c_consumer(force=True)
    assert os.path.exists(out)
    import subprocess
    needed = subprocess.run(["readelf", "-d", out], capture_output=True, text=True).stdout
    assert "libgconv_adapter.so" in needed and "libtorch" not in needed and "libpython" not in needed
    assert "libstdc++" not in needed        # (a C program: the C++ runtime comes in only through the library itself)


def test_cpu_quota_parsing_and_the_host_core_budget():
    """Round 5: an N-rank run sizes each rank's pinned core set from the container's CFS
    quota, and its bench line carries the quota and the throttle counters. The parsers, on the texts the kernel
    writes (cgroup v2 `cpu.max` / `cpu.stat`, v1 `cpu.stat`)."""
    d = ga.dist
    assert d.parse_cpu_max("1600000 100000\n") == 16.0
    assert d.parse_cpu_max("150000 100000") == 1.5
    assert d.parse_cpu_max("max 100000\n") is None
    assert d.parse_cpu_max("") is None and d.parse_cpu_max("garbage") is None and d.parse_cpu_max("0 100000") is None
    v2 = "usage_usec 123\nuser_usec 100\nsystem_usec 23\nnr_periods 420\nnr_throttled 7\nthrottled_usec 81234\nnr_bursts 0\n"
    assert d.parse_cpu_stat(v2) == {"nr_periods": 420, "nr_throttled": 7, "throttled_usec": 81234}
    v1 = "nr_periods 10\nnr_throttled 2\nthrottled_time 5000000\n"
    assert d.parse_cpu_stat(v1) == {"nr_periods": 10, "nr_throttled": 2, "throttled_usec": 5000}
    assert d.parse_cpu_stat("") == {"nr_periods": 0, "nr_throttled": 0, "throttled_usec": 0}
    # 8 ranks under the 16-CPU quota of a GPU box: 2 cores each (8 x 3 busy threads would be 1.5 x the share)
    assert d.host_core_budget(8, 16.0) == 2
    assert d.host_core_budget(4, 16.0) == 4 and d.host_core_budget(2, 16.0) == 8 and d.host_core_budget(1, 16.0) == 8
    assert d.host_core_budget(8, 4.0) == 1                      # never below one core
    assert d.host_core_budget(8, 256.0) == 8                    # never above the CCD's worth
    assert isinstance(d.host_core_budget(8), int)               # reads this container's own quota
    stats = d.cpu_throttle_stats()
    assert set(stats) == {"nr_periods", "nr_throttled", "throttled_usec"}
    q = d.cpu_quota()
    assert q is None or q > 0


def test_zero_grad_helper_sets_every_gradient_to_none():
    import torch
    ps = [torch.nn.Parameter(torch.randn(3)) for _ in range(4)]
    for p in ps[:3]:
        p.grad = torch.ones(3)
    ga.optim.zero_grad(ps)
    assert all(p.grad is None for p in ps)
    ps[0].grad = torch.ones(3)
    ga.optim.zero_grad(iter(ps[:1]), set_to_none=False)
    assert torch.equal(ps[0].grad, torch.zeros(3))
