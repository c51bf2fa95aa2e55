# Round-5 evidence passes, one GPU call:
#   A  MFMA utilisation of the HEADLINE kernels (tile_fwd / tile_bwd on hiv4096 and the 512-molecule shard)
#   B  where their waves wait (SQ wait / issue buckets)
#   C  the general SpMM on the banded control: why L2-resident gathers stop at ~9 TB/s
#   usage (GPU box, repo root): bash tools/lab/r05_evidence.sh
OUT=gpurun_out/r05_ev
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$ROOT/$OUT/counters.txt" 2>&1
echo "[A] tile MFMA pass"; date
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/tile_mfma" -- python3 "$ROOT/tools/tilebench.py" hiv4096 hiv > "$ROOT/$OUT/tile_mfma.log" 2>&1
echo "[B] tile wait pass"; date
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/tile_wait" -- python3 "$ROOT/tools/tilebench.py" hiv4096 hiv > "$ROOT/$OUT/tile_wait.log" 2>&1
echo "[C] banded SpMM passes"; date
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/banded_sq" -- python3 "$ROOT/tools/spmm_only.py" --banded > "$ROOT/$OUT/banded_sq.log" 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/banded_tcp" -- python3 "$ROOT/tools/spmm_only.py" --banded > "$ROOT/$OUT/banded_tcp.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$ROOT/$OUT/banded_tcc" -- python3 "$ROOT/tools/spmm_only.py" --banded > "$ROOT/$OUT/banded_tcc.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/arxiv_sq" -- python3 "$ROOT/tools/spmm_only.py" > "$ROOT/$OUT/arxiv_sq.log" 2>&1
cd "$ROOT"
echo "[D] reduce"; date
python3 tools/lab/pmc_reduce.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep the per-kernel reductions, drop the raw CSVs (tens of MB)
find "$OUT" -name "*counter_collection.csv" -size +4M -delete
echo "[E] baseline bench"; date
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 3000 "$OUT/bench.json"
