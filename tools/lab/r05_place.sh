OUT=gpurun_out/r05_place; rm -rf "$OUT"; mkdir -p "$OUT"; ROOT=$(pwd)
for rep in 1 2; do
  for tree in r04 r05; do
    if [ $tree = r04 ]; then cd "$ROOT/_old"; else cd "$ROOT"; fi
    echo -n "$tree cached: " >> "$ROOT/$OUT/out.txt"
    GCONV_LAB_STATIC=1 GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
    echo -n "$tree fresh: " >> "$ROOT/$OUT/out.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
    echo -n "$tree esol fresh: " >> "$ROOT/$OUT/out.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 esol 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
  done
done
cd "$ROOT"; cat "$OUT/out.txt"
