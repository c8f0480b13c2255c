set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "four_phase" > gpurun_out/r3_c14_tests.log 2>&1; echo "tests rc=$?"; grep -n "assert\|Error\|passed\|failed" gpurun_out/r3_c14_tests.log | head -20
timeout -k 10 300 python bench.py --workload config2 --steps 3 --warmup 1 --check --no-cpu-baseline > gpurun_out/r3_c14_bench_c2.json 2> gpurun_out/r3_c14_bench_c2.err; echo "rc=$?"
