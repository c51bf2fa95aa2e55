OUT=gpurun_out/r05_bisect; rm -rf "$OUT"; mkdir -p "$OUT"; ROOT=$(pwd)
for rep in 1 2; do
  cd "$ROOT/_old"; echo "== r04" >> "$ROOT/$OUT/out.txt"
  timeout -k 10 300 python3 tools/tilebench.py hiv esol hiv4096 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
  echo -n "r04 hiv fresh: " >> "$ROOT/$OUT/out.txt"
  GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
  cd "$ROOT"
  echo "== r05" >> "$OUT/out.txt"
  timeout -k 10 300 python3 tools/tilebench.py hiv esol hiv4096 2>/dev/null >> "$OUT/out.txt" || exit 1
  echo -n "r05 hiv fresh: " >> "$ROOT/$OUT/out.txt"
  GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
done
cat "$OUT/out.txt"
