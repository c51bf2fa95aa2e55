# main-stream timeline of the whole-batch step, kernel by kernel (duration and the gap in front of it), fresh plan per step against one cached plan,
# and which side-stream (graph build) kernels run beside each position
OUT=gpurun_out/r05_timeline; rm -rf "$OUT"; mkdir -p "$OUT"; ROOT=$(pwd)
timeout -k 10 600 python3 -m pytest tests/test_gpu_eval.py -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }; tail -1 "$OUT/pytest.log"
for mode in ${MODES:-fresh fresh_recorded rotate cached}; do
  unset GCONV_LAB_ROTATE GCONV_LAB_STATIC GCONV_NO_COMPLETION_EVENT
  if [ $mode = rotate ]; then export GCONV_LAB_ROTATE=1 GCONV_LAB_STATIC=1; fi
  if [ $mode = fresh_recorded ]; then export GCONV_NO_COMPLETION_EVENT=1; fi
  if [ $mode = cached ]; then export GCONV_LAB_STATIC=1; fi
  unset GCONV_LAB_NO_DISCARD
  if [ $mode = fresh_nodiscard ]; then export GCONV_LAB_NO_DISCARD=1; fi
  (cd /tmp && export TMPDIR=/tmp GCONV_LAB_SUMMARY=1 && rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/$mode" -- python3 "$ROOT/tools/lab/settle_trace.py" 2 hiv4096 > "$ROOT/$OUT/run_$mode.log" 2>&1) || exit 1
  python3 - "$OUT/$mode" $mode >> "$OUT/out.txt" <<'PY'
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
def short(r): return r["Kernel_Name"].split("(")[0].replace("void gconv::", "").replace("gconv::", "")[:44]
main = [r for r in rows if "tile_fwd" in r["Kernel_Name"] or "tile_bwd" in r["Kernel_Name"] or "finalize" in r["Kernel_Name"]]
side = [r for r in rows if r not in main] if len(rows) < 200000 else []
# cut into steps at the finalize kernel
steps, cur = [], []
for r in main:
    cur.append(r)
    if "finalize" in r["Kernel_Name"]:
        steps.append(cur); cur = []
steps = [s for s in steps if len(s) == 11][1:-1]
print(f"== {sys.argv[2]}: {len(steps)} steps of 11 main-stream kernels")
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for s in steps:
    for i, r in enumerate(s):
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        dur[i].append(en - st)
        if prev_end is not None: gap[i].append(st - prev_end)
        prev_end = en
beside = collections.defaultdict(collections.Counter)
si = 0
side_iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r)) for r in side]
for s in steps[:400]:
    for i, r in enumerate(s):
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        for a, b, nm in side_iv:
            if a < en and b > st: beside[i][nm] += min(b, en) - max(a, st)
tot = 0
for i in range(11):
    d = sorted(dur[i]); g = sorted(gap[i]) or [0]
    tot += d[len(d) // 2] + g[len(g) // 2]
    b = ", ".join(f"{nm} {ns / len(steps[:400]) / 1e3:.1f}us" for nm, ns in beside[i].most_common(3))
    print(f"  {i:2d} {short(steps[0][i]):44s} dur median {d[len(d) // 2] / 1e3:7.2f} us  gap in front p10 {g[len(g) // 10] / 1e3:5.2f} median {g[len(g) // 2] / 1e3:5.2f} p90 {g[len(g) * 9 // 10] / 1e3:5.2f} us   beside: {b}")
print(f"  sum of medians {tot / 1e3:.1f} us")
PY
  rm -rf "$OUT/$mode"
done
cat "$OUT/out.txt"
# and unprofiled medians of the same three
for mode in ${MODES2:-fresh fresh_recorded rotate fresh fresh_recorded rotate cached}; do
  unset GCONV_LAB_ROTATE GCONV_LAB_STATIC GCONV_NO_COMPLETION_EVENT
  if [ $mode = rotate ]; then export GCONV_LAB_ROTATE=1 GCONV_LAB_STATIC=1; fi
  if [ $mode = fresh_recorded ]; then export GCONV_NO_COMPLETION_EVENT=1; fi
  if [ $mode = cached ]; then export GCONV_LAB_STATIC=1; fi
  echo -n "$mode: " >> "$OUT/step.txt"
  GCONV_LAB_SUMMARY=1 timeout -k 10 300 python3 tools/lab/settle_trace.py 4 hiv4096 2>/dev/null >> "$OUT/step.txt" || exit 1
done
cat "$OUT/step.txt"
