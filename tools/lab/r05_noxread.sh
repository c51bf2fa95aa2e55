# Lab: upper bound of keeping the forward's X tile on chip for phase 4 (lab library built with
# -DGCONV_LAB_FWD_NO_XREAD: phase 4 adds zeros instead of re-reading X; results wrong by construction).
OUT=gpurun_out/r05_noxread; rm -rf "$OUT"; mkdir -p "$OUT"
for lib in product noxread; do
  if [ $lib = noxread ]; then export GCONV_ADAPTER_LIB=$(pwd)/gconv-adapter_amd/lib/libgconv_adapter_noxread.so; else unset GCONV_ADAPTER_LIB; fi
  echo "== $lib" >> "$OUT/tilebench.txt"
  timeout -k 10 300 python3 tools/tilebench.py hiv4096 hiv esol >> "$OUT/tilebench.txt" 2>&1 || exit 1
done
grep -v amdgpu "$OUT/tilebench.txt"
