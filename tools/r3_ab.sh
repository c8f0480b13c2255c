cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix" 2>&1 | tail -2
for rep in 1 2 3; do
for v in v4 new; do
  if [ $v = v4 ]; then export MM_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libmm_v4.so; else unset MM_LIB_PATH; fi
  timeout -k 10 300 python bench.py --precision matrix --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), round(d['roofline']['dominant_launch']['avg_ms'],3))"
done
done
unset MM_LIB_PATH
timeout -k 10 300 python bench.py --workload config2 --steps 3 --warmup 1 --check --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['check']); print({k:[v2 for k2,v2 in d[k].items() if 'identical' in k2] for k in ('fast_screen','bounded_search','f64_exact')})"
