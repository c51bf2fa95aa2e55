# Kernel timeline of the headline step (bench.py without extras / CPU baseline), with a fresh plan per step and with
# one cached plan: where the fresh plan's cost sits — longer tile kernels (contention with the build's kernels) or
# gaps between them.   usage (GPU box, repo root): bash tools/lab/step_timeline.sh ["--workload hiv --scaling strong"]
ARGS=${1:---workload hiv --scaling strong}
ROOT=$(pwd)
for mode in fresh cached; do
  rm -rf $ROOT/gpurun_out/step_tl_$mode
  ( cd /tmp && export TMPDIR=/tmp && if [ $mode = cached ]; then export GCONV_BENCH_STATIC_PLAN=1; fi; \
    rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/step_tl_$mode -- python3 $ROOT/bench.py --no-extra --no-cpu-baseline --steps 100 --warmup 10 $ARGS > $ROOT/gpurun_out/step_tl_$mode.json 2> $ROOT/gpurun_out/step_tl_$mode.err )
done
python3 - <<'PY'
import csv, glob, json
for mode in ("fresh", "cached"):
    f = glob.glob(f"gpurun_out/step_tl_{mode}/*/*kernel_trace.csv")[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    main = [r for r in rows if "tile_fwd_kernel" in r[2] or "tile_bwd_kernel" in r[2] or "finalize_batched" in r[2]]
    main = main[-11 * 60:]                      # the last 60 steps (5 fwd + 5 bwd + 1 finalize each)
    t_lo, t_hi = main[0][0], main[-1][1]
    fwd = [(e - s) / 1e3 for s, e, n in main if "tile_fwd" in n]
    bwd = [(e - s) / 1e3 for s, e, n in main if "tile_bwd" in n]
    gaps = [(main[i + 1][0] - main[i][1]) / 1e3 for i in range(len(main) - 1)]
    others = [(s, e, n) for s, e, n in rows if t_lo <= s <= t_hi and not any(k in n for k in ("tile_fwd_kernel", "tile_bwd_kernel", "finalize_batched"))]
    busy = sum(e - s for s, e, n in others) / 1e3
    line = json.loads(open(f"gpurun_out/step_tl_{mode}.json").read().strip().splitlines()[-1])
    print(f"{mode:6s}: step {line['ms_per_step']:.4f} ms under the profiler | per step: tile_fwd {sum(fwd) / 60:.1f} us (avg {sum(fwd) / len(fwd):.1f}), "
          f"tile_bwd {sum(bwd) / 60:.1f} (avg {sum(bwd) / len(bwd):.1f}), gaps between the step's kernels {sum(gaps) / 60:.1f}, "
          f"other kernels in the window (build, sort, ...) {busy / 60:.1f} us of kernel time in {len(others) / 60:.1f} launches")
PY
rm -rf gpurun_out/step_tl_fresh gpurun_out/step_tl_cached
