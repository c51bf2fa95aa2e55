# counting builder: symmetry test per ROW (product) against per EDGE of the edge list (lab library = the previous build)
OUT=gpurun_out/r05_rowsym; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_eval.py tests/test_gpu_robustness.py tests/test_gpu_locality.py -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -20 "$OUT/pytest.log"; exit 1; }
tail -1 "$OUT/pytest.log"
for rep in 1 2 3; do
  for lib in edgesym product; do
    if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
    echo -n "$lib fresh: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
for sh in 0/2 0/8; do
for lib in edgesym product; do
  if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
  echo -n "shard $sh $lib: " >> "$OUT/step.txt"
  GCONV_LAB_SHARD=$sh GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 3 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
done
done
unset GCONV_ADAPTER_LIB
python3 bench.py --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); e = d['extra']; print('hiv512 build wall us', e['hiv512_graph_build_wall_us'], 'esol', e['esol32_graph_build_wall_us'], 'headline', d['ms_per_step'], 'hiv512', e['hiv512_ms_per_step'])" >> "$OUT/step.txt"
cat "$OUT/step.txt"
MODES='fresh' MODES2='rotate' bash tools/lab/r05_timeline.sh | grep -v "^   [1-4] \|^  10\|^   9"
