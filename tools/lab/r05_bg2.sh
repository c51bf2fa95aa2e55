OUT=gpurun_out/r05_bg2; rm -rf "$OUT"; mkdir -p "$OUT"
for rep in 1 2; do
for bg in default wsplit fused; do
  if [ $bg = default ]; then unset GCONV_BG_BUILD; else export GCONV_BG_BUILD=$bg; fi
  echo -n "bg=$bg hiv4096: " >> "$OUT/step.txt"
  GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
done
done
unset GCONV_BG_BUILD
echo -n "cached: " >> "$OUT/step.txt"
GCONV_LAB_STATIC=1 GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$OUT/step.txt"
cat "$OUT/step.txt"
