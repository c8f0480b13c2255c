cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in 0.005 0.0002; do
  MM_BENCH_SWITCH_INTERVAL=$v MM_BENCH_REHEARSE_WORLD=8 timeout -k 10 300 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('switch $v', round(d['per_rank_ms_per_step'],3), round(d['dominant_launch_ms'],3))"
done
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N1', round(d['ms_per_step'],3))"
