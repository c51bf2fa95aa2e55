# Lab: the prefetch worker's builder next to the whole-batch step (GCONV_BG_BUILD: radix default | split | wide-split | fused)
OUT=gpurun_out/r05_bg; rm -rf "$OUT"; mkdir -p "$OUT"
for bg in default split wsplit fused; do
  if [ $bg = default ]; then unset GCONV_BG_BUILD; else export GCONV_BG_BUILD=$bg; fi
  for w in hiv4096 hiv; do
    echo -n "bg=$bg $w fresh: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 $w 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
unset GCONV_BG_BUILD
export GCONV_LAB_STATIC=1
echo -n "cached hiv4096: " >> "$OUT/step.txt"
GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
cat "$OUT/step.txt"
