OUT=gpurun_out/r05_prio2; rm -rf "$OUT"; mkdir -p "$OUT"
for sh in 0/2 0/4 0/8; do
  export GCONV_LAB_SHARD=$sh
  for lib in product prioall; do
    if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
    for bg in radix split; do
      export GCONV_BG_BUILD=$bg
      echo -n "shard $sh $lib bg=$bg: " >> "$OUT/step.txt"
      GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
    done
  done
done
cat "$OUT/step.txt"
