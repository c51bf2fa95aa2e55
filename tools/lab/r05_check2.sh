# full GPU suite at HEAD, then under the knobs that change the prefetch / binding path
OUT=gpurun_out/r05_check2; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
tail -1 "$OUT/pytest.log"
for knob in GCONV_PY_BINDING GCONV_NO_TILE_PATH GCONV_NO_COUNTING_BUILD; do
  env $knob=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/$knob.log" 2>&1 || { tail -30 "$OUT/$knob.log"; exit 1; }
  echo "$knob: $(tail -1 $OUT/$knob.log)"
done
python3 bench.py --steps 200 --warmup 20 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
tail -1 "$OUT/bench.json" | cut -c1-300
