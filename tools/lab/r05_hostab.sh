OUT=gpurun_out/r05_hostab; rm -rf "$OUT"; mkdir -p "$OUT"; ROOT=$(pwd)
for rep in 1 2; do
  for tree in r04 r05; do
    if [ $tree = r04 ]; then cd "$ROOT/_old"; else cd "$ROOT"; fi
    for mode in fresh cached; do
      echo "== $tree hiv $mode" >> "$ROOT/$OUT/out.txt"
      timeout -k 10 300 python3 tools/stepprof.py hiv $mode 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
    done
  done
done
cd "$ROOT"; cat "$OUT/out.txt"
