"""Reduces the rocprofv3 --pmc CSVs written by tools/pmc_traffic.sh to one JSON: per-launch averages of
FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS for the SpMM kernel and the derived traffic
(2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies a 128-B read request as 64 B — MI355X_MICROARCH.md, HBM)."""
import csv, glob, json, os, sys

out = sys.argv[1]
vals = {}
for path in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if "spmm_rows_kernel" not in row["Kernel_Name"]:
                continue
            vals.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            vals[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
avg = {k: {"avg_per_launch": sum(v.values()) / len(v), "launches": len(v)} for k, v in vals.items()}
fetch_raw = avg["FETCH_SIZE"]["avg_per_launch"] * 1024     # KiB
write = avg["WRITE_SIZE"]["avg_per_launch"] * 1024
hit, miss = avg["TCC_HIT_sum"]["avg_per_launch"], avg["TCC_MISS_sum"]["avg_per_launch"]
res = {"workload": "S-arxiv N=170000 nnz=1199998 F=32 normalized, production flags (prescaled in/out)",
       "command": "bash tools/pmc_traffic.sh  (rocprofv3 --pmc <group> --output-format csv -- python3 tools/spmm_only.py; "
                  "one pass per group: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum)",
       "kernels": {"spmm_rows_kernel<32,norm,GENERIC>": avg},
       "derived": {"commit": os.environ.get("GCONV_COMMIT", "n/a"), "fetch_bytes_raw": fetch_raw, "fetch_bytes_x2_gfx950": 2 * fetch_raw, "write_bytes": write,
                   "traffic_bytes_per_launch": 2 * fetch_raw + write, "l2_hit_rate": hit / (hit + miss),
                   "tcc_miss_x128B": miss * 128,
                   "note": "FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 tallies a 128-B read request as 64 B "
                           "(MI355X_MICROARCH.md), hence x2 on FETCH_SIZE; TCC_MISS_sum x 128 B is the cross-check"}}
with open(os.path.join(out, "spmm_arxiv_pmc.json"), "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps(res["derived"], indent=1))
