# Same-box A/B of the round-5 tree against the round-4 tree (a git worktree of f6580e8 under _old/, built here):
# the shard, the whole batch (fresh / cached plan) and ESOL, interleaved, medians over 4 s each.
OUT=gpurun_out/r05_vs_r04; rm -rf "$OUT"; mkdir -p "$OUT"
ROOT=$(pwd)
for rep in 1; do
  for tree in r04 r05; do
    if [ $tree = r04 ]; then cd "$ROOT/_old"; else cd "$ROOT"; fi
    for w in hiv hiv4096 esol; do
      echo -n "$tree $w fresh: " >> "$ROOT/$OUT/step.txt"
      GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 $w 2>/dev/null >> "$ROOT/$OUT/step.txt" || exit 1
    done
    echo -n "$tree hiv4096 cached: " >> "$ROOT/$OUT/step.txt"
    GCONV_LAB_STATIC=1 GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$ROOT/$OUT/step.txt" || exit 1
  done
done
cd "$ROOT"; cat "$OUT/step.txt"
