set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_c9_gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3_c9_gpu_tests.log
tail -5 gpurun_out/r3_c9_gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_c9_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3_c9_smoke.log
