"""Per-step device totals of the fine-tune step from tools/lab/ft_prof.sh's kernel stats: python tools/lab/ft_kernels.py [min_us]"""
import csv, glob, sys
f = sorted(glob.glob("gpurun_out/ft_prof/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
steps = max(int(r["Calls"]) for r in rows if "adam_step" in r["Name"])
lo = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
tot = 0.0
for r in rows:
    if "flush_icache" in r["Name"]:
        continue
    per = float(r["TotalDurationNs"]) / steps / 1e3
    tot += per
    if per > lo:
        print("%7.1f us/step  calls/step %5.1f  avg %6.1f  %s" % (per, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:96]))
print("steps", steps, "sum per step", round(tot, 1), "us (under the profiler)")
