# full GPU suite + headline step traces (whole batch cached / fresh, shard) + the default bench line
OUT=gpurun_out/r05_check; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?
tail -3 "$OUT/pytest.log"
[ $rc -ne 0 ] && { grep -n "Error\|FAILED\|assert" "$OUT/pytest.log" | head -40; exit $rc; }
for plan in cached fresh; do
  if [ $plan = cached ]; then export GCONV_LAB_STATIC=1; else unset GCONV_LAB_STATIC; fi
  echo -n "hiv4096 $plan: " >> "$OUT/step.txt"
  GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 5 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
done
unset GCONV_LAB_STATIC
echo -n "hiv512 fresh: " >> "$OUT/step.txt"
GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv 2>/dev/null >> "$OUT/step.txt" || exit 1
cat "$OUT/step.txt"
timeout -k 10 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
tail -c 2500 "$OUT/bench.json"
