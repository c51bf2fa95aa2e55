# background split builder as 256 one-wave workgroups (product) against 64 four-wave workgroups (GCONV_BG_BUILD=split)
OUT=gpurun_out/r05_thin; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_eval.py tests/test_gpu_robustness.py -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -20 "$OUT/pytest.log"; exit 1; }
tail -1 "$OUT/pytest.log"
for rep in 1 2 3; do
  for bg in split default; do
    if [ $bg = default ]; then unset GCONV_BG_BUILD; else export GCONV_BG_BUILD=$bg; fi
    echo -n "bg=$bg fresh: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
for sh in 0/2 0/4; do
for bg in split default radix; do
  if [ $bg = default ]; then unset GCONV_BG_BUILD; else export GCONV_BG_BUILD=$bg; fi
  echo -n "shard $sh bg=$bg: " >> "$OUT/step.txt"
  GCONV_LAB_SHARD=$sh GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
done
done
unset GCONV_BG_BUILD
cat "$OUT/step.txt"
MODES='fresh' MODES2='rotate' bash tools/lab/r05_timeline.sh | grep -v "^   [1-4] \|^  10\|^   9"
