for i in 1 2; do
python3 bench.py --scaling weak --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fresh ', d['ms_per_step'], d['config']['workload'][:40])"
GCONV_BENCH_STATIC_PLAN=1 python3 bench.py --scaling weak --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('static', d['ms_per_step'], d['config']['workload'][:40])"
done
