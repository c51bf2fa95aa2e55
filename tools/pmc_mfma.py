"""Reduces the rocprofv3 --pmc CSV of tools/pmc_mfma.sh: per kernel, average SQ_VALU_MFMA_BUSY_CYCLES,
SQ_BUSY_CYCLES and their ratio (the share of the shader engines' busy time during which a matrix
instruction was executing)."""
import csv, glob, json, os, sys

out = sys.argv[1]
vals = {}
for path in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0]
            if "gconv" not in name:
                continue
            d = vals.setdefault(name, {}).setdefault(row["Counter_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
res = {"workload": "S-arxiv adapter fwd + bwd (N=170000, d=128, rank 32, normalized)", "commit": os.environ.get("GCONV_COMMIT", "n/a"),
       "command": "bash tools/pmc_mfma.sh (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES)", "kernels": {}}
for name, ctr in sorted(vals.items()):
    avg = {k: sum(v.values()) / len(v) for k, v in ctr.items()}
    avg["launches"] = len(next(iter(ctr.values())))
    if avg.get("SQ_BUSY_CYCLES"):
        avg["mfma_busy_over_sq_busy"] = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / avg["SQ_BUSY_CYCLES"]
    res["kernels"][name] = avg
with open(os.path.join(out, "arxiv_mfma_pmc.json"), "w") as fh:
    json.dump(res, fh, indent=1)
for k, v in res["kernels"].items():
    print(f"{k[:70]:70s} mfma/busy {v.get('mfma_busy_over_sq_busy', 0):.3f}")
