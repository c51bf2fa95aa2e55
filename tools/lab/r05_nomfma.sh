# Lab: upper bound of any cheaper matrix arithmetic — a library whose fp32 MFMAs are ONE fma each
# (-DGCONV_LAB_NO_MFMA; results wrong by construction), against the product, per kernel under rocprofv3 --kernel-trace --stats
OUT=gpurun_out/r05_nomfma; rm -rf "$OUT"; mkdir -p "$OUT"
ROOT=$(pwd)
for lib in product nomfma; do
  if [ $lib = product ]; then unset GCONV_ADAPTER_LIB; else export GCONV_ADAPTER_LIB=$ROOT/gconv-adapter_amd/lib/libgconv_adapter_$lib.so; fi
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$lib" -- python3 "$ROOT/tools/arxiv_adapter_only.py" > "$ROOT/$OUT/$lib.log" 2>&1) || exit 1
  cp "$OUT/$lib"/*/*kernel_stats.csv "$OUT/${lib}_kernel_stats.csv"; rm -rf "$OUT/$lib"
  echo "== $lib tilebench" >> "$OUT/tilebench.txt"
  timeout -k 10 300 python3 tools/tilebench.py hiv4096 hiv 2>/dev/null >> "$OUT/tilebench.txt" || exit 1
done
python3 - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
t = {}
for lib in ("product", "nomfma"):
    for r in csv.DictReader(open(f"{out}/{lib}_kernel_stats.csv")):
        n = r["Name"].split("(")[0].replace("void ", "")
        if "gconv" in n and int(r["Calls"]) >= 10:
            t.setdefault(n, {})[lib] = float(r["AverageNs"]) / 1e3
for n, v in sorted(t.items(), key=lambda kv: -kv[1].get("product", 0)):
    if "product" in v and "nomfma" in v:
        print(f"{n[:60]:60s} product {v['product']:7.1f} us   no-MFMA {v['nomfma']:7.1f} us   -{100 * (1 - v['nomfma'] / v['product']):4.1f} %")
PY
cat "$OUT/tilebench.txt"
