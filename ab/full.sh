#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_gpu_suite.txt 2>&1
rc=$?; echo "suite rc=$rc" >> gpurun_out/r3b_gpu_suite.txt; tail -3 gpurun_out/r3b_gpu_suite.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/soak_matrix.py 300 > gpurun_out/r3b_matrix_soak.txt 2>&1
rc=$?; echo "soak rc=$rc" >> gpurun_out/r3b_matrix_soak.txt; tail -4 gpurun_out/r3b_matrix_soak.txt
exit $rc
