"""`python bench.py --gpus N` with no torchrun around it (the driver's command form): the process becomes a
launcher, the N ranks are its children, rank 0's JSON line is the launcher's last stdout line. Rehearsed here
with 2 ranks on the one GPU of the test box over gloo (`GCONV_BENCH_BACKEND=gloo`: several ranks per GPU are
refused otherwise) — the collective path, the strong-scaling shards and the reporting, not a measurement."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_two_ranks_started_by_bench_itself(dev):
    out = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--no-extra", "--no-cpu-baseline"],
               {"GCONV_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "strong"
    cfg = line["config"]
    assert cfg["world_size"] == 2 and cfg["backend"] == "gloo" and len(cfg["devices"]) == 2
    assert "2048 graphs/GPU" in cfg["workload"] and "gloo" in cfg["workload"] and "RCCL" not in cfg["workload"]
    assert line["value"] > 0 and line["unit"] == "graphs/s" and line["vs_baseline"] is None
    assert "roofline" not in line and "cpu_baseline" not in line        # N = 1 only


def test_more_ranks_than_gpus_is_refused_without_the_rehearsal_switch(dev):
    import torch
    n = torch.cuda.device_count() + 1
    out = _run(["--gpus", str(n), "--steps", "2", "--no-extra", "--no-cpu-baseline"], {"GCONV_BENCH_BACKEND": ""})
    assert out.returncode != 0 and "GPU(s) visible" in out.stderr
    assert not out.stdout.strip().startswith("{")


def test_graph_replay_child_line(dev):
    """`bench.py --graph-replay` (the child process of the default run): configs[3]'s adapter step eager and as ONE
    replayed HIP graph, for both graphs, as one JSON line; the replay is not slower than driving the same launches
    from the host."""
    out = _run(["--graph-replay"], {}, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for name in ("cora", "pubmed"):
        eager, replay = line[f"{name}_adapter_step_eager_us"], line[f"{name}_adapter_step_hip_graph_us"]
        assert 0 < replay <= eager * 1.25, (name, eager, replay)   # (a sanity bound, not a performance gate)
