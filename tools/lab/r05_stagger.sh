# Lab: walk order of a persistent workgroup's tiles (GCONV_TILE_STAGGER: 0 common order, 1 odd workgroups reversed,
# k > 1: k groups of workgroups with rotated starts) — the headline step with one cached plan and with a fresh plan.
OUT=gpurun_out/r05_stagger; rm -rf "$OUT"; mkdir -p "$OUT"
true
true
for st in ${MODES:-0 1 3 5 7}; do
  export GCONV_TILE_STAGGER=$st
  for plan in cached fresh; do
    if [ $plan = cached ]; then export GCONV_LAB_STATIC=1; else unset GCONV_LAB_STATIC; fi
    echo -n "stagger=$st $plan: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 5 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
  done
done
cat "$OUT/step.txt"
