# Round-end measurement set: the driver's bench line, the same command under rocprofv3 (kernel trace +
# stats), and a 2-rank rehearsal of the N > 1 path on this one GPU (gloo, both ranks on device 0).
#   usage (GPU box, repo root): bash tools/lab/final_bench.sh gpurun_out/final
OUT=${1:-gpurun_out/final}
ROOT=$(pwd)
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
tail -1 "$OUT/bench.json"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof" -- python3 "$ROOT/bench.py" > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/bench_under_rocprof.err") || exit 1
cp "$OUT"/prof/*/*kernel_stats.csv "$OUT/kernel_stats.csv"
rm -rf "$OUT/prof"   # (the raw kernel trace is tens of MB: only the per-kernel summary is kept)
# the driver's command form for N > 1: bench.py starts its ranks itself (gloo: both ranks on this one GPU)
GCONV_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 > "$OUT/bench_2rank_gloo.json" 2> "$OUT/bench_2rank_gloo.err" || exit 1
tail -1 "$OUT/bench_2rank_gloo.json" | cut -c1-600
GCONV_COMMIT=${GCONV_COMMIT:-n/a} bash tools/pmc_mfma.sh "$OUT/pmc_mfma" > "$OUT/pmc_mfma.txt" 2>&1 || exit 1
GCONV_COMMIT=${GCONV_COMMIT:-n/a} bash tools/pmc_traffic.sh "$OUT/pmc_traffic" > "$OUT/pmc_traffic.txt" 2>&1 || exit 1
rm -rf "$OUT"/pmc_mfma/sq "$OUT"/pmc_traffic/FETCH_SIZE "$OUT"/pmc_traffic/WRITE_SIZE "$OUT"/pmc_traffic/TCC_HIT_sum_TCC_MISS_sum
