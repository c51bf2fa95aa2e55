cd /tmp && export TMPDIR=/tmp
for v in full noload nomfma nothing; do
  if [ $v = full ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$GRAFT_REPO_ROOT/gconv-adapter_amd/lib/libgconv_adapter_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mlp_$v -- python3 $GRAFT_REPO_ROOT/tools/lab/mlp_bench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_mlp_$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for v in ("full", "noload", "nomfma", "nothing"):
    f = glob.glob(f"gpurun_out/prof_mlp_{v}/*/*kernel_stats.csv")[0]
    print(v)
    for row in csv.DictReader(open(f)):
        if "gconv" in row["Name"]:
            print(f"   {row['Name'][:58]:58s} calls {row['Calls']:>4s} avg {float(row['AverageNs'])/1e3:8.1f} us")
PY
