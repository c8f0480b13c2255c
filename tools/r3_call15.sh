set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python tools/diag_mx.py 2>&1 | grep -v amdgpu.ids | tail -5
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "four_phase or config4 or ladder" > gpurun_out/r3_c15_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3_c15_tests.log
timeout -k 10 300 python bench.py --workload config2 --steps 3 --warmup 1 --check --no-cpu-baseline > gpurun_out/r3_c15_bench_c2.json 2> gpurun_out/r3_c15_bench_c2.err; echo "rc=$?"
