# kernel-trace statistics of the shard step with ONE cached plan, both trees (where do +3 us per step sit?)
OUT=gpurun_out/r05_trace_ab; rm -rf "$OUT"; mkdir -p "$OUT"; ROOT=$(pwd)
for tree in r04 r05; do
  if [ $tree = r04 ]; then T="$ROOT/_old"; else T="$ROOT"; fi
  (cd /tmp && export TMPDIR=/tmp GCONV_LAB_STATIC=1 GCONV_LAB_SUMMARY=1 && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$tree" -- python3 "$T/tools/lab/settle_trace.py" 2 hiv > "$ROOT/$OUT/$tree.log" 2>&1) || exit 1
  cp "$OUT/$tree"/*/*kernel_stats.csv "$OUT/${tree}_kernel_stats.csv"
  python3 - "$OUT/$tree" "$tree" >> "$OUT/out.txt" <<'PY'
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]          # steady state
import collections
dur = collections.defaultdict(list); gaps = []
for a, b in zip(rows, rows[1:]):
    dur[a["Kernel_Name"].split("(")[0][-60:]].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
steps = sum(1 for r in rows if "finalize" in r["Kernel_Name"])
print(f"{sys.argv[2]}: {steps} steps, {span / steps / 1e3:.2f} us/step; sum of gaps per step {sum(gaps) / steps / 1e3:.2f} us")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"   {k:60s} n/step {len(v) / steps:5.2f}  avg {sum(v) / len(v) / 1e3:7.2f} us  per step {sum(v) / steps / 1e3:7.2f} us")
PY
  rm -rf "$OUT/$tree"
done
cat "$OUT/out.txt"
