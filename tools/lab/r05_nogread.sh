# Lab: bound of removing the backward's phase-4 re-read of G (-DGCONV_LAB_BWD_NO_GREAD: G = 0 there, wrong results):
# the kernel alone (tilebench) and the whole-batch / shard steps, product against lab library, same box.
ROOT=$(pwd)
for lib in product nogread product nogread; do
  if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$ROOT/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
  echo "== $lib"
  timeout -k 10 200 python3 tools/tilebench.py hiv4096 hiv 2>/dev/null | tail -4
  python3 bench.py --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step hiv4096', d['ms_per_step'])"
  GCONV_BENCH_STATIC_PLAN=1 python3 bench.py --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step hiv4096 cached plan', d['ms_per_step'])"
done
