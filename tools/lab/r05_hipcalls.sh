# which HIP runtime calls the fresh-plan step makes per step (against the cached-plan step)
OUT=gpurun_out/r05_hipcalls; rm -rf "$OUT"; mkdir -p "$OUT"; ROOT=$(pwd)
for mode in fresh cached; do
  if [ $mode = cached ]; then export GCONV_LAB_STATIC=1; else unset GCONV_LAB_STATIC; fi
  (cd /tmp && export TMPDIR=/tmp GCONV_LAB_SUMMARY=1 && rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$mode" -- python3 "$ROOT/tools/lab/settle_trace.py" 1 hiv4096 > "$ROOT/$OUT/run_$mode.log" 2>&1) || exit 1
  ls "$OUT/$mode"/*/ >> "$OUT/out.txt"
  python3 - "$OUT/$mode" $mode >> "$OUT/out.txt" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
steps = sum(1 for r in csv.DictReader(open(kt)) if "finalize" in r["Kernel_Name"])
print(f"== {sys.argv[2]}: {steps} steps")
for f in glob.glob(d + "/*/*hip_api_stats.csv") + glob.glob(d + "/*/*hip_runtime_api_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(f"   {r['Name']:40s} calls/step {int(r['Calls']) / steps:7.2f}  avg {float(r['AverageNs']) / 1e3:8.2f} us")
PY
  rm -rf "$OUT/$mode"
done
cat "$OUT/out.txt"
