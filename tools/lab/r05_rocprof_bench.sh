# The driver's bench command under rocprofv3 --kernel-trace --stats alone (the middle part of final_bench.sh)
OUT=${1:-gpurun_out/final}
ROOT=$(pwd)
mkdir -p "$OUT"; rm -rf "$OUT/prof"
(cd /tmp && export TMPDIR=/tmp GCONV_BENCH_NO_GRAPH=1 && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof" -- python3 "$ROOT/bench.py" > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/bench_under_rocprof.err") || exit 1
cp "$(ls -S "$OUT"/prof/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv" || exit 1
rm -rf "$OUT/prof"
tail -1 "$OUT/bench_under_rocprof.json" | cut -c1-200
head -5 "$OUT/kernel_stats.csv"
