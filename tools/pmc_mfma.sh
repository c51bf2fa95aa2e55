#!/bin/bash
# MFMA utilisation of the S-arxiv adapter's kernels: ONE rocprofv3 PMC pass (SQ counters only, never
# combined with trace domains), reduced by tools/pmc_mfma.py to profiles/rNN_arxiv_mfma_pmc.json.
#   usage (on the GPU box, from the repo root):  bash tools/pmc_mfma.sh gpurun_out/pmc_mfma
set -e
OUT=${1:-gpurun_out/pmc_mfma}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/sq" -- python3 "$ROOT/tools/arxiv_adapter_only.py" > "$ROOT/$OUT/sq.log" 2>&1
cd "$ROOT"
python3 tools/pmc_mfma.py "$OUT"
