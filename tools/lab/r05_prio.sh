OUT=gpurun_out/r05_prio; rm -rf "$OUT"; mkdir -p "$OUT"
for rep in 1 2; do
for lib in product prio; do
  if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
  for w in hiv4096 hiv; do
    echo -n "$lib $w fresh: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 $w 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
  echo -n "$lib hiv4096 cached: " >> "$OUT/step.txt"
  GCONV_LAB_STATIC=1 GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
done
done
cat "$OUT/step.txt"
