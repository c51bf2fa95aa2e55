"""Reduces the rocprofv3 --pmc CSV of tools/pmc_mfma.sh: per kernel, average SQ_VALU_MFMA_BUSY_CYCLES
(summed over the chip's 1024 SIMDs: 32 cycles per 16x16x4 fp32 MFMA, 8 per 4x4x1) and GRBM_GUI_ACTIVE
(summed over the 8 XCDs: / 8 = the kernel's length in shader clocks, MI355X_MICROARCH.md "DVFS").
  mfma_util = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8): the share of the matrix pipes' time in use.
(Round 2 divided by SQ_BUSY_CYCLES, which counts per shader engine and gave ratios above 1.)"""
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
       "command": "bash tools/pmc_mfma.sh (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE)", "kernels": {}}
for name, ctr in sorted(vals.items()):
    avg = {k: sum(v.values()) / len(v) for k, v in ctr.items()}
    avg["launches"] = len(next(iter(ctr.values())))
    if avg.get("GRBM_GUI_ACTIVE"):
        avg["kernel_cycles"] = avg["GRBM_GUI_ACTIVE"] / 8
        avg["mfma_util"] = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * avg["kernel_cycles"])
    res["kernels"][name] = avg
with open(os.path.join(out, "arxiv_mfma_pmc.json"), "w") as fh:
    json.dump(res, fh, indent=1)
for k, v in res["kernels"].items():
    print(f"{k[:70]:70s} mfma util {v.get('mfma_util', 0):.3f}  cycles {v.get('kernel_cycles', 0):.0f}")
