# per-kernel device times of the 512-molecule adapter step (bench.py --workload hiv --scaling weak) under rocprofv3
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/shard_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/shard_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload hiv --scaling weak --no-cpu-baseline --no-extra --steps 300 > $GRAFT_REPO_ROOT/gpurun_out/shard_prof.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/shard_prof/*/*kernel_trace.csv   # (tens of MB)
tail -1 gpurun_out/shard_prof.log | cut -c1-160
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/shard_prof/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"   {r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  {float(r['Percentage']):5.1f}%")
PY
