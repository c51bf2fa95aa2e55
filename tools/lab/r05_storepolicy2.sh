# Lab: output-store cache policy in the step, all workloads: product (nt everywhere) | st0 (tile kernels default policy)
# | st00 (tile kernels AND the general path's K3 / B4 outputs default policy)
OUT=gpurun_out/r05_storepolicy2; rm -rf "$OUT"; mkdir -p "$OUT"
for lib in product st0 st00; do
  if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
  for w in arxiv esol hiv; do
    echo -n "$lib $w: " >> "$OUT/step.txt"
    timeout -k 10 300 python3 bench.py --workload $w --scaling weak --no-cpu-baseline --no-extra --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> "$OUT/step.txt" || exit 1
  done
  echo -n "$lib tilebench: " >> "$OUT/step.txt"
  timeout -k 10 300 python3 tools/tilebench.py hiv4096 hiv esol 2>/dev/null | tr '\n' '|' >> "$OUT/step.txt"; echo >> "$OUT/step.txt"
done
cat "$OUT/step.txt"
