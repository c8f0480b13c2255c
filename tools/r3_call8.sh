set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 5 120 tools/bin/ubench_mfma16 > gpurun_out/r3_ubench_mfma16_b.txt 2>&1; tail -9 gpurun_out/r3_ubench_mfma16_b.txt
timeout -k 10 300 python tools/bench_api.py --stages --repeats 9 > gpurun_out/r3_c8_api_stages.txt 2>&1; grep -v amdgpu.ids gpurun_out/r3_c8_api_stages.txt | tail -24
timeout -k 10 600 python -m pytest tests/test_golden_and_api.py tests/test_gpu_property.py tests/test_gpu_workflow.py -x -q -m gpu > gpurun_out/r3_c8_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3_c8_tests.log
tail -3 gpurun_out/r3_c8_tests.log
