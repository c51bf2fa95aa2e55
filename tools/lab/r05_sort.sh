# the tile-list sort as a 4-wave workgroup (runs beside a backward tile workgroup) against the 16-wave one (lab library = the previous build)
OUT=gpurun_out/r05_sort; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_eval.py -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -20 "$OUT/pytest.log"; exit 1; }
tail -1 "$OUT/pytest.log"
for rep in 1 2 3; do
  for lib in oldsort product; do
    if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
    echo -n "$lib fresh: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
unset GCONV_ADAPTER_LIB
echo -n "product cached: " >> "$OUT/step.txt"
GCONV_LAB_STATIC=1 GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt"
for sh in 0/2; do
  for lib in oldsort product; do
    if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
    echo -n "shard $sh $lib: " >> "$OUT/step.txt"
    GCONV_LAB_SHARD=$sh GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
cat "$OUT/step.txt"
