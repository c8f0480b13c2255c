set -x
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix" > gpurun_out/r3_c11_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3_c11_tests.log
timeout -k 10 300 python bench.py --precision matrix --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline > gpurun_out/r3_c11_bench_matrix.json 2> gpurun_out/r3_c11_bench_matrix.err; echo "bench rc=$?"
cd /tmp
O=$R/gpurun_out/r3_mxpmc; mkdir -p $O
B="python3 $R/bench.py --precision matrix --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs"
for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$tag -- $B > $O/$tag.json 2> $O/$tag.err; echo "pmc $tag rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/r3_mxpmc > gpurun_out/r3_c11_mx_pmc_summary.csv
grep "k_screen_mx" gpurun_out/r3_c11_mx_pmc_summary.csv | grep ",47617024,"
rm -rf gpurun_out/r3_mxpmc
