# K0: cooperative launch of the one-launch builder, and what the prefetch worker's build
# costs the step under each background flavour (radix pipeline | one narrow launch per phase | narrow fused launch)
echo "== in-stream wall time per plan: plain launch / cooperative launch"
python tools/lab/k0_time.py
GCONV_COUNT_COOPERATIVE=1 python tools/lab/k0_time.py
for bg in radix split fused; do echo "== background flavour $bg"; GCONV_BG_BUILD=$bg python tools/lab/k0_time.py | grep -v arxiv; done
B="python bench.py --no-extra --no-cpu-baseline --steps 400"
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_step', d['ms_per_step'], 'graphs/s', d['value'])"; }
for wl in "--workload hiv --scaling weak" "--workload hiv --scaling strong" "--workload esol"; do
  echo "== step, $wl: cached plan (no build beside the step)"; GCONV_BENCH_STATIC_PLAN=1 $B $wl | one
  for bg in radix split fused; do echo "== step, $wl: fresh plan per step, background build = $bg"; GCONV_BG_BUILD=$bg $B $wl | one; done
done
