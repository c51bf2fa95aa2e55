# HBM-side traffic of the fused tile kernels (round 4): FETCH_SIZE / WRITE_SIZE / L2 hit per launch, one rocprofv3 --pmc
# pass per counter group over tools/tilebench.py; is X (forward) / G (backward) fetched past L2 once or twice?
#   usage (GPU box, repo root): bash tools/lab/tile_traffic.sh hiv4096
W=${1:-hiv4096}
OUT=gpurun_out/tile_traffic_$W
ROOT=$(pwd)
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo "$grp" | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/$tag" -- python3 "$ROOT/tools/tilebench.py" $W > "$ROOT/$OUT/$tag.log" 2>&1
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
vals = {}
for path in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "tile_fwd" not in k and "tile_bwd" not in k:
            continue
        name = "fwd" if "tile_fwd" in k else "bwd"
        vals.setdefault((name, row["Counter_Name"]), {}).setdefault(row["Dispatch_Id"], 0.0)
        vals[(name, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
import json
res = {"workload": os.path.basename(out).replace("tile_traffic_", ""), "commit": os.environ.get("GCONV_COMMIT", "n/a"),
       "command": "bash tools/lab/tile_traffic.sh <workload>  (rocprofv3 --pmc <group> -- python3 tools/tilebench.py <workload>; one "
                  "pass per group: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum; gfx950 x2 correction on FETCH_SIZE)"}
for name in ("fwd", "bwd"):
    a = {c: sum(v.values()) / len(v) for (n, c), v in vals.items() if n == name}
    f, w = a.get("FETCH_SIZE", 0) * 1024 * 2, a.get("WRITE_SIZE", 0) * 1024
    hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 1)
    res[f"tile_{name}"] = {"fetch_bytes_x2_gfx950": f, "write_bytes": w, "traffic_bytes_per_launch": f + w,
                           "l2_hit_rate": hit / (hit + miss), "tcc_miss_x128B": miss * 128}
    print(f"tile_{name}: fetched past L2 (2 x FETCH_SIZE) {f / 1e6:.1f} MB, written {w / 1e6:.1f} MB, "
          f"traffic {(f + w) / 1e6:.1f} MB per launch; L2 hit rate {hit / (hit + miss):.3f}")
json.dump(res, open(os.path.join(out, "tile_traffic.json"), "w"), indent=1)
PY
