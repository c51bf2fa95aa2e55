OUT=gpurun_out/r05_stamps; rm -rf "$OUT"; mkdir -p "$OUT"; ROOT=$(pwd)
for tree in r04 r05 r04 r05; do
  if [ $tree = r04 ]; then cd "$ROOT/_old"; else cd "$ROOT"; fi
  echo "== $tree" >> "$ROOT/$OUT/out.txt"
  timeout -k 10 300 python3 tools/tilestamps.py hiv 2>/dev/null >> "$ROOT/$OUT/out.txt" || exit 1
done
cd "$ROOT"; cat "$OUT/out.txt"
