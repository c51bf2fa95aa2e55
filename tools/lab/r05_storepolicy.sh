# Lab: cache policy of the tile kernels' output stores IN THE STEP (out_k is the next kernel's X): nt (product, aux 2)
# against the default policy (lab library, -DGCONV_STORE_AUX=0). Round 1 chose nt on the kernel alone.
OUT=gpurun_out/r05_storepolicy; rm -rf "$OUT"; mkdir -p "$OUT"
for lib in product st0; do
  if [ $lib = st0 ]; then export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_st0.so; else unset GCONV_ADAPTER_LIB; fi
  for plan in cached fresh; do
    if [ $plan = cached ]; then export GCONV_LAB_STATIC=1; else unset GCONV_LAB_STATIC; fi
    echo -n "$lib $plan: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 5 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
  unset GCONV_LAB_STATIC
  echo -n "$lib hiv512 fresh: " >> "$OUT/step.txt"
  GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv 2>/dev/null >> "$OUT/step.txt" || exit 1
done
cat "$OUT/step.txt"
