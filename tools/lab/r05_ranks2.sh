OUT=gpurun_out/r05_ranks2; rm -rf "$OUT"; mkdir -p "$OUT"
run() {  # n cores tag
  if [ "$2" = auto ]; then unset GCONV_CORES_PER_RANK; else export GCONV_CORES_PER_RANK=$2; fi
  GCONV_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus $1 --steps 100 --warmup 10 > "$OUT/bench_$1rank_$2.json" 2> "$OUT/bench_$1rank_$2.err" || { tail -5 "$OUT/bench_$1rank_$2.err"; return 1; }
  tail -1 "$OUT/bench_$1rank_$2.json" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['diagnostics']; print('ranks', d['n_gpus'], 'cores/rank', d['config'].get('cores_per_rank'), 'ms/step', d['ms_per_step'], 'allreduce wall us', g['all_reduce']['wall_us_median_max_over_ranks'], 'cfs', g['cfs_throttle_delta_over_timed_region'], g['pinned_cores_per_rank'])"
}
run 4 2 || exit 1
run 4 4 || exit 1
run 4 8 || exit 1
run 5 auto || exit 1
run 5 8 || exit 1
GCONV_NO_PIN=1 run 4 auto
