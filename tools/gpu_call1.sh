#!/bin/bash
# first GPU call: VALU microbench, parity tests, first bench line, rocprof kernel trace
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $O/device.txt
nproc >> $O/device.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $O/device.txt
echo "== ubench"; timeout -k 10 120 ./tools/ubench_valu > $O/ubench.txt 2>&1; rc=$?; echo "ubench rc=$rc"; [ $rc -ge 124 ] && exit $rc
echo "== pytest"; timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 $O/pytest_gpu.log; [ $rc -ge 124 ] && exit $rc
echo "== smoke"; timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -3 $O/smoke.log; [ $rc -ge 124 ] && exit $rc
echo "== bench"; timeout -k 10 600 python bench.py --steps 2 --warmup 1 > $O/bench_config2.json 2> $O/bench_config2.err; rc=$?; echo "bench rc=$rc"; cat $O/bench_config2.json; tail -5 $O/bench_config2.err; [ $rc -ge 124 ] && exit $rc
echo "== rocprof"; cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/rocprof_bench.json 2> $O/rocprof.err; rc=$?; echo "rocprof rc=$rc"; tail -3 $O/rocprof.err
find $O/prof_r1 -name "*stats*" | head
exit 0
