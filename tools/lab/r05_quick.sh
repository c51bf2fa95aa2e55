OUT=gpurun_out/r05_quick; rm -rf "$OUT"; mkdir -p "$OUT"
for w in esol hiv hiv4096; do
  echo -n "$w: " >> "$OUT/step.txt"
  GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 $w 2>/dev/null >> "$OUT/step.txt" || exit 1
done
cat "$OUT/step.txt"
