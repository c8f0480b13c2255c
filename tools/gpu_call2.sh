#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest"; timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest_gpu.log; [ $rc -ge 124 ] && exit $rc
for cfg in "config2 decoupled" "config2 chain" "config3 decoupled"; do
  set -- $cfg
  echo "== bench $1 $2"; timeout -k 10 600 python bench.py --steps 3 --warmup 1 --workload $1 --mode $2 > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; rc=$?; echo "rc=$rc"; cat $O/bench_$1_$2.json; tail -3 $O/bench_$1_$2.err; [ $rc -ge 124 ] && exit $rc
done
echo "== cpu thread sweep"; for t in 8 16 32 64; do timeout -k 10 200 python bench.py --steps 1 --warmup 0 --workload tiny --cpu-threads $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($t, d['cpu_baseline'])"; done
echo "== rocprof"; cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1b -- python3 $R/bench.py --steps 2 --warmup 1 --workload config3 --no-cpu-baseline > $O/rocprof_bench.json 2> $O/rocprof.err; rc=$?; echo "rocprof rc=$rc"; tail -2 $O/rocprof.err
exit 0
