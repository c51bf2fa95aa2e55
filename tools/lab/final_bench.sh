# Round-end measurement set: the GPU suite, the driver's bench line, the same command under rocprofv3 (kernel trace +
# stats), 2- and 4-rank rehearsals of the N > 1 path on this one GPU (gloo), and the PMC passes the line cites.
#   usage (GPU box, repo root): bash tools/lab/final_bench.sh gpurun_out/final
OUT=${1:-gpurun_out/final}
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
tail -1 "$OUT/bench.json" | cut -c1-400
if [ -z "$SKIP_ROCPROF" ]; then
# (GCONV_BENCH_NO_GRAPH: the HIP-graph replay figures come from a CHILD process of bench.py — under the profiler that is a
#  second trace directory; the summary kept here is the one process that runs the headline and the roofline kernel)
(cd /tmp && export TMPDIR=/tmp GCONV_BENCH_NO_GRAPH=1 && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof" -- python3 "$ROOT/bench.py" > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/bench_under_rocprof.err") || exit 1
cp "$(ls -S "$OUT"/prof/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv" || exit 1
fi
rm -rf "$OUT/prof"   # (the raw kernel trace is tens of MB: only the per-kernel summary is kept)
# the driver's command form for N > 1: bench.py starts its ranks itself (gloo: the ranks share this one GPU)
for n in 2 4; do
  GCONV_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus $n --steps 100 --warmup 10 > "$OUT/bench_${n}rank_gloo.json" 2> "$OUT/bench_${n}rank_gloo.err" || exit 1
  tail -1 "$OUT/bench_${n}rank_gloo.json" | cut -c1-300
done
GCONV_COMMIT=${GCONV_COMMIT:-n/a} bash tools/pmc_mfma.sh "$OUT/pmc_mfma" > "$OUT/pmc_mfma.txt" 2>&1 || exit 1
GCONV_COMMIT=${GCONV_COMMIT:-n/a} bash tools/pmc_traffic.sh "$OUT/pmc_traffic" > "$OUT/pmc_traffic.txt" 2>&1 || exit 1
GCONV_COMMIT=${GCONV_COMMIT:-n/a} bash tools/lab/tile_traffic.sh hiv4096 > "$OUT/tile_traffic.txt" 2>&1 || exit 1
cp gpurun_out/tile_traffic_hiv4096/tile_traffic.json "$OUT/tile_traffic_hiv4096.json"
rm -rf "$OUT"/pmc_mfma/sq "$OUT"/pmc_traffic/FETCH_SIZE "$OUT"/pmc_traffic/WRITE_SIZE "$OUT"/pmc_traffic/TCC_HIT_sum_TCC_MISS_sum gpurun_out/tile_traffic_hiv4096
tail -3 "$OUT/tile_traffic.txt"
