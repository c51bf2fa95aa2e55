# N > 1 rehearsal on the one-GPU box (gloo: ranks wrap onto the one card, at most 6 processes may use it): the
# self-diagnosing line of VERDICT r04 item 3 — per-rank step times, the all-reduce alone, CFS quota and throttle counters
OUT=gpurun_out/r05_ranks; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_readout.py tests/test_gpu_bench_launcher.py -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
cat /sys/fs/cgroup/cpu.max > "$OUT/cpu_max.txt" 2>/dev/null; cat /sys/fs/cgroup/cpu.stat > "$OUT/cpu_stat_before.txt" 2>/dev/null
for n in 2 4 6; do
  GCONV_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus $n --steps 100 --warmup 10 > "$OUT/bench_${n}rank_gloo.json" 2> "$OUT/bench_${n}rank_gloo.err" || { tail -20 "$OUT/bench_${n}rank_gloo.err"; exit 1; }
  tail -1 "$OUT/bench_${n}rank_gloo.json" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['ms_per_step'], d['config'].get('cpu_quota'), d['config'].get('cores_per_rank'), json.dumps(d.get('diagnostics'))[:900])"
done
# the same 6 ranks with 8 cores each (what rounds 1-4 pinned): does the quota throttle them?
GCONV_CORES_PER_RANK=8 GCONV_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 6 --steps 100 --warmup 10 > "$OUT/bench_6rank_gloo_8cores.json" 2> "$OUT/bench_6rank_gloo_8cores.err" || exit 1
tail -1 "$OUT/bench_6rank_gloo_8cores.json" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('8 cores/rank:', d['n_gpus'], d['ms_per_step'], d['config'].get('cores_per_rank'), json.dumps(d.get('diagnostics'))[:900])"
cat /sys/fs/cgroup/cpu.stat > "$OUT/cpu_stat_after.txt" 2>/dev/null
