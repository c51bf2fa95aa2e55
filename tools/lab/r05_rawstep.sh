# every kernel of three consecutive fresh-plan steps of the whole batch with its queue and its start / end relative to the first
OUT=gpurun_out/r05_rawstep; rm -rf "$OUT"; mkdir -p "$OUT"; ROOT=$(pwd)
(cd /tmp && export TMPDIR=/tmp GCONV_LAB_SUMMARY=1 && rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/t" -- python3 "$ROOT/tools/lab/settle_trace.py" 1 hiv4096 > "$ROOT/$OUT/run.log" 2>&1) || exit 1
python3 - "$OUT/t" > "$OUT/out.txt" <<'PY'
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
print(list(rows[0].keys()))
rows = rows[len(rows) * 3 // 4:]
i0 = next(i for i, r in enumerate(rows) if "finalize" in r["Kernel_Name"])
t0 = int(rows[i0]["Start_Timestamp"])
n = 0
for r in rows[i0:]:
    nm = r["Kernel_Name"].split("(")[0].replace("void gconv::", "").replace("gconv::", "")[:46]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.2f} .. {(int(r['End_Timestamp']) - t0) / 1e3:9.2f}  q{r.get('Queue_Id', '?'):>3s} {nm}")
    n += "finalize" in r["Kernel_Name"]
    if n == 4: break
PY
rm -rf "$OUT/t"; cat "$OUT/out.txt"
