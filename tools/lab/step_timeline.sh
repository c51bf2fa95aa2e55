# kernel timeline of the headline step (bench.py without extras / CPU baseline): per-kernel gaps
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/step_tl -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/step_tl.json 2> $GRAFT_REPO_ROOT/gpurun_out/step_tl.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/step_tl/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
# the last 3 steps: find the tile_fwd<8,19,true> kernels
idx = [i for i, r in enumerate(rows) if "tile_fwd_kernel<8, 19, true>" in r[2]]
start = idx[-15]     # 3 steps x 5 fwd
t0 = rows[start][0]
prev_end = None
for s, e, n, q in rows[start:start + 60]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:6.1f}  gap {gap:6.1f}  q{q} {n}")
    prev_end = max(prev_end or 0, e)
PY
rm -rf gpurun_out/step_tl
