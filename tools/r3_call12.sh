set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix" > gpurun_out/r3_c12_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3_c12_tests.log
for nb in 2 1; do
MM_MX_NB=$nb timeout -k 10 300 python bench.py --precision matrix --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline > gpurun_out/r3_c12_bench_matrix_nb$nb.json 2> gpurun_out/r3_c12_bench_matrix.err; echo "bench nb=$nb rc=$?"
done
