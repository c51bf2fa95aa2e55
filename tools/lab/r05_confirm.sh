# Quick confirmation of a tree on one box: the GPU suite, smoke(), the driver's bench line.
#   usage (GPU box, repo root): bash tools/lab/r05_confirm.sh gpurun_out/confirm
OUT=${1:-gpurun_out/confirm}
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { tail -20 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
tail -1 "$OUT/bench.json" | cut -c1-300
