OUT=gpurun_out/r05_alt; rm -rf "$OUT"; mkdir -p "$OUT"
for alt in 0 1 0 1; do
  if [ $alt = 1 ]; then export GCONV_TILE_ALTERNATE=1; else unset GCONV_TILE_ALTERNATE; fi
  for plan in cached fresh; do
    if [ $plan = cached ]; then export GCONV_LAB_STATIC=1; else unset GCONV_LAB_STATIC; fi
    echo -n "alternate=$alt $plan: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
cat "$OUT/step.txt"
