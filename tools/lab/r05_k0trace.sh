# which kernels the fresh-plan step of the whole batch runs per step (main + side stream), and for how long
OUT=gpurun_out/r05_k0trace; rm -rf "$OUT"; mkdir -p "$OUT"; ROOT=$(pwd)
(cd /tmp && export TMPDIR=/tmp GCONV_LAB_SUMMARY=1 && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/t" -- python3 "$ROOT/tools/lab/settle_trace.py" 2 hiv4096 > "$ROOT/$OUT/run.log" 2>&1) || exit 1
python3 - "$OUT/t" > "$OUT/out.txt" <<'PY'
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
steps = sum(1 for r in rows if "finalize" in r["Kernel_Name"])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{steps} steps, {span / steps / 1e3:.1f} us/step (profiled)")
dur = collections.defaultdict(list)
for r in rows:
    dur[(r["Kernel_Name"].split("(")[0][-70:], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"   {k[0]:70s} grid {k[1]:>8s} wg {k[2]:>5s}  n/step {len(v) / steps:5.2f}  avg {sum(v) / len(v) / 1e3:7.2f} us  per step {sum(v) / steps / 1e3:7.2f} us")
PY
rm -rf "$OUT/t"; cat "$OUT/out.txt"
