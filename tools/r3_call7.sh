set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_golden_and_api.py tests/test_gpu_property.py tests/test_gpu_workflow.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r3_c7_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3_c7_tests.log
tail -5 gpurun_out/r3_c7_tests.log
timeout -k 10 300 python tools/bench_api.py --stages --repeats 9 > gpurun_out/r3_c7_api_stages.txt 2>&1; tail -32 gpurun_out/r3_c7_api_stages.txt
timeout -k 10 300 python tools/bench_published.py > gpurun_out/r3_c7_published.txt 2>&1; tail -5 gpurun_out/r3_c7_published.txt
