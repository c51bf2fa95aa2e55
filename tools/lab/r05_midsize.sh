# the N = 2 and N = 4 points of the strong-scaling curve on ONE GPU (2048 / 1024 molecules): walk order with two passes
OUT=gpurun_out/r05_midsize2; rm -rf "$OUT"; mkdir -p "$OUT"
for sh in 0/2 0/4; do
  export GCONV_LAB_SHARD=$sh
  for st in 1 2 1 2; do
    export GCONV_TILE_STAGGER=$st
    echo -n "shard $sh stagger=$st: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
unset GCONV_TILE_STAGGER GCONV_LAB_SHARD
echo -n "hiv512: " >> "$OUT/step.txt"; GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv 2>/dev/null >> "$OUT/step.txt"
echo -n "hiv4096: " >> "$OUT/step.txt"; GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt"
cat "$OUT/step.txt"
