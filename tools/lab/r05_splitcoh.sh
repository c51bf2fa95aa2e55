# the split background builder with plain cached accesses in its phase-B launch (product) against device-scope accesses (lab library)
OUT=gpurun_out/r05_splitcoh; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_eval.py tests/test_gpu_robustness.py -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -20 "$OUT/pytest.log"; exit 1; }
tail -1 "$OUT/pytest.log"
for rep in 1 2 3; do
  for lib in splitcoh product; do
    if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
    echo -n "$lib fresh: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
for lib in splitcoh product; do
  if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
  echo -n "shard 0/2 $lib: " >> "$OUT/step.txt"
  GCONV_LAB_SHARD=0/2 GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
done
unset GCONV_ADAPTER_LIB
cat "$OUT/step.txt"
MODES='fresh' MODES2='rotate' bash tools/lab/r05_timeline.sh | grep -v "^   [1-4] \|^  10\|^   9"
