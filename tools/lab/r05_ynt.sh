# the SpMM's output rows (H2', P, dH1', T: read by the NEXT kernel of the adapter) non-temporal (product, round 4's choice on the
# kernel alone) against default policy (lab library -DGCONV_SPMM_Y_NT=0): the S-arxiv step and the kernel alone
OUT=gpurun_out/r05_ynt; rm -rf "$OUT"; mkdir -p "$OUT"
for rep in 1 2; do
for lib in product ynt0; do
  if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
  echo -n "$lib arxiv step ms: " >> "$OUT/out.txt"
  timeout -k 10 300 python3 bench.py --workload arxiv --no-cpu-baseline --no-extra --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> "$OUT/out.txt" || exit 1
  echo -n "$lib " >> "$OUT/out.txt"
  timeout -k 10 300 python3 tools/spmm_only.py --time 2>/dev/null | tail -1 >> "$OUT/out.txt" || exit 1
  for w in cora pubmed; do
    echo -n "$lib $w step ms: " >> "$OUT/out.txt"
    timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --no-extra --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> "$OUT/out.txt" || exit 1
  done
done
done
cat "$OUT/out.txt"
