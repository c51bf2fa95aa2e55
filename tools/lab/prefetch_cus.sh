# The prefetch worker's graph build confined to k CUs (CU-masked side stream) while the tile kernels leave
# those CUs alone (round 4 lab): what a fresh plan per step then costs next to a cached plan.
B="python bench.py --no-extra --no-cpu-baseline --steps 400"
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_step', d['ms_per_step'], 'graphs/s', d['value'])"; }
for wl in "--workload hiv --scaling weak" "--workload hiv --scaling strong" "--workload esol"; do
  echo "== $wl: cached plan, 256 CUs"; GCONV_BENCH_STATIC_PLAN=1 $B $wl | one
  echo "== $wl: fresh plan, product (radix pipeline on an unmasked side stream)"; $B $wl | one
  for k in 8 16; do
    t=$((256 - k))
    echo "== $wl: cached plan, tile kernels on $t CUs"; GCONV_TILE_CUS=$t GCONV_BENCH_STATIC_PLAN=1 $B $wl | one
    for bg in split fused radix; do
      echo "== $wl: fresh plan, build = $bg on $k CUs, tile kernels on $t"; GCONV_PREFETCH_CUS=$k GCONV_TILE_CUS=$t GCONV_BG_BUILD=$bg $B $wl | one
    done
  done
done
