# the GPU suite under the A/B knobs DESIGN §9 says it passes under (KNOBS="..." selects)
OUT=gpurun_out/r05_knobs; mkdir -p "$OUT"
for knob in ${KNOBS:-GCONV_NO_TILE_PATH GCONV_PY_BINDING GCONV_AUTOGRAD_PARAM_GRADS GCONV_NO_FUSED_GATHER GCONV_NO_SMALL_BUILD GCONV_NO_COUNTING_BUILD}; do
  echo "== $knob=1" | tee -a "$OUT/summary.txt"
  env $knob=1 timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > "$OUT/$knob.log" 2>&1
  tail -3 "$OUT/$knob.log" | tee -a "$OUT/summary.txt"
done
