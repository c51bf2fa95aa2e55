# where the small-batch regression of the round-5 tree comes from: kernels alone (tilebench), the shard step with
# a cached plan (no graph build), with a fresh one
OUT=gpurun_out/r05_vs_r04b; rm -rf "$OUT"; mkdir -p "$OUT"
ROOT=$(pwd)
for rep in 1; do
  for tree in r04 r05; do
    if [ $tree = r04 ]; then cd "$ROOT/_old"; else cd "$ROOT"; fi
    echo "== $tree" >> "$ROOT/$OUT/out.txt"
    timeout -k 10 300 python3 tools/tilebench.py hiv esol 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
    for w in hiv esol; do
      echo -n "$tree $w cached: " >> "$ROOT/$OUT/out.txt"
      GCONV_LAB_STATIC=1 GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 $w 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
    done
  done
done
cd "$ROOT"; cat "$OUT/out.txt"
