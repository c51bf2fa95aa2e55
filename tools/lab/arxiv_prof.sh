# kernel trace of one S-arxiv adapter (forward + backward x 10): per-kernel average durations
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/arxiv_prof -- python3 $GRAFT_REPO_ROOT/tools/arxiv_adapter_only.py > $GRAFT_REPO_ROOT/gpurun_out/arxiv_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/arxiv_prof/*/*kernel_stats.csv")[0]
for row in csv.DictReader(open(f)):
    if "gconv" in row["Name"]:
        print(f"   {row['Name'][:70]:70s} calls {row['Calls']:>4s} avg {float(row['AverageNs'])/1e3:8.1f} us")
PY
