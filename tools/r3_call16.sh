set -x
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r3_c16_gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3_c16_gpu_tests.log
tail -5 gpurun_out/r3_c16_gpu_tests.log
