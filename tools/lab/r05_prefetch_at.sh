OUT=gpurun_out/r05_prefetch_at2; rm -rf "$OUT"; mkdir -p "$OUT"
for bg in default radix split; do
  if [ $bg = default ]; then unset GCONV_BG_BUILD; else export GCONV_BG_BUILD=$bg; fi
  for w in hiv4096 hiv esol; do
    echo -n "auto bg=$bg $w: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 $w 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
cat "$OUT/step.txt"
