# Lab: K1 / B1 (gemm_rowsA) memory-schedule variants at S-arxiv scale (LAB.md round 5, "K1 schedule")
OUT=gpurun_out/r05_k1; rm -rf "$OUT"; mkdir -p "$OUT"
ROOT=$(pwd)
for g in 0 443 532 665 886 1024; do
  LAB_TAG=grid$g GCONV_LAB_ROWSA_GRID=$g timeout -k 10 200 python3 tools/lab/r05_k1.py >> "$OUT/k1.txt" 2>"$OUT/err.txt" || { tail "$OUT/err.txt"; exit 1; }
done
for v in 1 2 4 8 3 5; do
  LAB_TAG=variant$v GCONV_ADAPTER_LIB=$ROOT/gconv-adapter_amd/lib/libgconv_adapter_k1v$v.so timeout -k 10 200 python3 tools/lab/r05_k1.py >> "$OUT/k1.txt" 2>"$OUT/err.txt" || { tail "$OUT/err.txt"; exit 1; }
done
cat "$OUT/k1.txt"
