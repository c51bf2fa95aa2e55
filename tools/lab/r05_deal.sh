# Round 5, VERDICT r04 item 2: the dynamic tile deal against the static one.
#   (1) the GPU suite on this build  (2) device time of the tile kernels alone, both deals
#   (3) the headline step with a fresh plan per step and with one cached plan, both deals (medians over 6 s)
#   usage (GPU box, repo root): bash tools/lab/r05_deal.sh
OUT=gpurun_out/r05_deal
rm -rf "$OUT"; mkdir -p "$OUT"
echo "[1] pytest -m gpu"; date
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?
tail -5 "$OUT/pytest.log"
[ $rc -ne 0 ] && { grep -n "Error\|FAILED\|assert" "$OUT/pytest.log" | head -40; exit $rc; }
echo "[2] tile kernels alone"; date
for deal in dynamic static; do
  if [ $deal = static ]; then export GCONV_TILE_STATIC_DEAL=1; else unset GCONV_TILE_STATIC_DEAL; fi
  echo "== $deal" >> "$OUT/tilebench.txt"
  timeout -k 10 300 python3 tools/tilebench.py hiv4096 hiv hiv:2048 esol >> "$OUT/tilebench.txt" 2>&1 || exit 1
done
cat "$OUT/tilebench.txt"
echo "[3] headline step"; date
for deal in dynamic static; do
  if [ $deal = static ]; then export GCONV_TILE_STATIC_DEAL=1; else unset GCONV_TILE_STATIC_DEAL; fi
  for plan in fresh cached; do
    if [ $plan = cached ]; then export GCONV_LAB_STATIC=1; else unset GCONV_LAB_STATIC; fi
    echo -n "$deal $plan: " >> "$OUT/step.txt"
    GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 6 hiv4096 >> "$OUT/step.txt" 2>&1 || exit 1
  done
done
unset GCONV_TILE_STATIC_DEAL GCONV_LAB_STATIC
cat "$OUT/step.txt"
date
echo "[4] SpMM row -> XCD schedule on the banded control (and S-arxiv, where it must stay off)"; date
for xr in 0 1 auto; do
  if [ $xr = auto ]; then unset GCONV_SPMM_XCD_ROWS; else export GCONV_SPMM_XCD_ROWS=$xr; fi
  echo "== GCONV_SPMM_XCD_ROWS=$xr" >> "$OUT/spmm_xcd.txt"
  timeout -k 10 300 python3 tools/spmm_only.py --banded --time >> "$OUT/spmm_xcd.txt" 2>&1 || exit 1
  timeout -k 10 300 python3 tools/spmm_only.py --time >> "$OUT/spmm_xcd.txt" 2>&1 || exit 1
done
unset GCONV_SPMM_XCD_ROWS
cat "$OUT/spmm_xcd.txt"
date
