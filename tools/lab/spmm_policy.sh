# S-arxiv rank-32 normalized SpMM under each cache-policy build (tools/lab/spmm_policy.py builds them):
# launch time (HIP events, best of 5 x 50) for every variant, then the traffic PMC passes for the product
# build and for the variants named in $PMC (default: ycolseg).
L=gconv-adapter_amd/lib
echo "product: $(python3 tools/spmm_only.py --time)"
for v in y col seg ycol ycolseg; do
  echo "$v: $(GCONV_ADAPTER_LIB=$L/libgconv_adapter_spmm_$v.so python3 tools/spmm_only.py --time)"
done
echo "product again: $(python3 tools/spmm_only.py --time)"
export GCONV_COMMIT=$(cat gpurun_out/.commit 2>/dev/null || echo n/a)
bash tools/pmc_traffic.sh gpurun_out/pmc_r04_product > /dev/null && cp gpurun_out/pmc_r04_product/spmm_arxiv_pmc.json gpurun_out/r04_spmm_arxiv_pmc_product.json
for v in ${PMC:-ycolseg}; do
  GCONV_ADAPTER_LIB=$(pwd)/$L/libgconv_adapter_spmm_$v.so bash tools/pmc_traffic.sh gpurun_out/pmc_r04_$v > /dev/null && cp gpurun_out/pmc_r04_$v/spmm_arxiv_pmc.json gpurun_out/r04_spmm_arxiv_pmc_$v.json
done
python3 - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r04_spmm_arxiv_pmc_*.json")):
    d = json.load(open(p))["derived"]
    print(p, "traffic MB %.1f  fetch x2 MB %.1f  write MB %.1f  L2 hit %.3f" % (d["traffic_bytes_per_launch"] / 1e6, d["fetch_bytes_x2_gfx950"] / 1e6, d["write_bytes"] / 1e6, d["l2_hit_rate"]))
PY
