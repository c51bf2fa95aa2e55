# per-kernel durations of tools/lab/mlp_bench.py (13104 rows) under rocprofv3, W-stationary and tiled GEMMs
cd /tmp && export TMPDIR=/tmp
for ws in 1 0; do
  export GCONV_MLP_WS=$ws
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mlp_ws$ws
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mlp_ws$ws -- python3 $GRAFT_REPO_ROOT/tools/lab/mlp_bench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_mlp_ws$ws.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for v in ("ws1", "ws0"):
    f = glob.glob(f"gpurun_out/prof_mlp_{v}/*/*kernel_stats.csv")[0]
    print(v)
    for row in csv.DictReader(open(f)):
        if "gconv" in row["Name"] and ("f16" in row["Name"] or "aggregate" in row["Name"]):
            print(f"   {row['Name'][:100]:100s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:8.1f} us")
PY
