# whole-batch step under other tile sizes (tiles per persistent workgroup k = 2 .. 6), fresh plan per step and one cached plan
OUT=gpurun_out/r05_tsweep; rm -rf "$OUT"; mkdir -p "$OUT"
for tm in 128 140 208 90 72; do
  for mode in fresh cached; do
    if [ $mode = cached ]; then export GCONV_LAB_STATIC=1; else unset GCONV_LAB_STATIC; fi
    echo -n "TMAX=$tm $mode: " >> "$OUT/step.txt"
    GCONV_TILE_NODES_MAX=$tm GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
  GCONV_TILE_NODES_MAX=$tm timeout -k 10 300 python3 tools/tilebench.py hiv4096 2>&1 | grep -i "hiv\|tile" | tail -3 >> "$OUT/step.txt"
done
cat "$OUT/step.txt"
