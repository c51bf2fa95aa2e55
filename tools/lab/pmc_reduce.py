"""Per-kernel averages of every counter in the rocprofv3 --pmc CSVs below a directory (one sub-directory per pass):
   python3 tools/lab/pmc_reduce.py gpurun_out/r05_ev  ->  <dir>/pmc_summary.json + a table on stdout.
Derived: kernel_cycles = GRBM_GUI_ACTIVE / 8 (sum over the 8 XCDs, MI355X_MICROARCH.md "DVFS");
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel_cycles); the SQ wait buckets as shares of SQ_WAVE_CYCLES."""
import csv, glob, json, os, sys

root = sys.argv[1]
res = {}
for pas in sorted(os.listdir(root)):
    pdir = os.path.join(root, pas)
    if not os.path.isdir(pdir):
        continue
    vals = {}
    for path in glob.glob(os.path.join(pdir, "*", "*counter_collection.csv")) + glob.glob(os.path.join(pdir, "*", "*", "*counter_collection.csv")):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if "gconv" not in name:
                    continue
                # the same kernel on two workloads of one pass (tilebench hiv4096 hiv): told apart by its launch shape
                name = f"{name} grid={row['Grid_Size']} lds={row['LDS_Block_Size']}"
                d = vals.setdefault(name, {}).setdefault(row["Counter_Name"], {})
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    out = {}
    for name, ctr in sorted(vals.items()):
        avg = {k: sum(v.values()) / len(v) for k, v in ctr.items()}
        avg["launches"] = len(next(iter(ctr.values())))
        if avg.get("GRBM_GUI_ACTIVE"):
            avg["kernel_cycles"] = avg["GRBM_GUI_ACTIVE"] / 8
            if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                avg["mfma_util"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * avg["kernel_cycles"])
        wc = avg.get("SQ_WAVE_CYCLES")
        if wc:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
                if k in avg:
                    avg[k + "_share_of_wave_cycles"] = avg[k] / wc
        out[name] = avg
    if out:
        res[pas] = out
json.dump(res, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
for pas, ks in res.items():
    print(f"== {pas}")
    for name, avg in ks.items():
        if avg["launches"] < 5:
            continue
        print(f"  {name[:96]:96s} n={avg['launches']:4d} " + " ".join(
            f"{k}={v:.4g}" for k, v in avg.items() if k not in ("launches",)))
