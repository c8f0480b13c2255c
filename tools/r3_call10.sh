set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix" > gpurun_out/r3_c10_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3_c10_tests.log
tail -30 gpurun_out/r3_c10_tests.log
timeout -k 10 300 python bench.py --precision matrix --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline > gpurun_out/r3_c10_bench_matrix.json 2> gpurun_out/r3_c10_bench_matrix.err; echo "bench rc=$?"; tail -3 gpurun_out/r3_c10_bench_matrix.err; cut -c1-300 gpurun_out/r3_c10_bench_matrix.json
