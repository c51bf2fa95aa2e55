# kernel trace of the fine-tune step (bench.measure_finetune_step): per-kernel totals per step
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ft_prof -- python3 $GRAFT_REPO_ROOT/tools/finetunestep.py 100 > $GRAFT_REPO_ROOT/gpurun_out/ft_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/ft_prof.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ft_prof/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over all steps; per step (/~125 incl. warm-up) {tot/1e3/125:.0f} us")
for r in rows[:32]:
    print(f"   {r['Name'][:84]:84s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.1f} us  {float(r['Percentage']):5.1f}%")
PY
