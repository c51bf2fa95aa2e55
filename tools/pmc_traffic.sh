#!/bin/bash
# Collects the HBM-side traffic of the S-arxiv SpMM with rocprofv3 PMC counters, ONE pass per counter
# group (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with trace domains), then
# tools/pmc_traffic.py reduces the CSVs to profiles/rNN_spmm_arxiv_pmc.json.
#   usage (on the GPU box, from the repo root):  bash tools/pmc_traffic.sh gpurun_out/pmc_r02
set -e
OUT=${1:-gpurun_out/pmc}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo "$grp" | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/$tag" -- python3 "$ROOT/tools/spmm_only.py" > "$ROOT/$OUT/$tag.log" 2>&1
done
cd "$ROOT"
python3 tools/pmc_traffic.py "$OUT"
