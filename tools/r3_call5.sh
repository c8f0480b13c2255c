set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/r3_c5_bench_default.json 2> gpurun_out/r3_c5_bench_default.err; echo "bench default rc=$?"
tail -3 gpurun_out/r3_c5_bench_default.err
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r3_c5_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3_c5_tests.log
tail -5 gpurun_out/r3_c5_tests.log
for n in 8 4 2; do
  MM_BENCH_REHEARSE_WORLD=$n timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c5_rehearse$n.json 2> gpurun_out/r3_c5_rehearse$n.err; echo "rehearse $n rc=$?"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/r3_c5_bench_n1.json 2> gpurun_out/r3_c5_bench_n1.err; echo "bench rc=$?"
bash tools/gpu_pmc.sh config3 > gpurun_out/r3_c5_pmc.log 2>&1; echo "pmc rc=$?"
python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/r3_config3_fast_pmc_summary.csv
f=$(find gpurun_out/pmc/trace -name "*kernel_stats.csv" | xargs ls -t | head -1); cp $f gpurun_out/r3_config3_default_kernel_stats.csv
f=$(find gpurun_out/pmc/trace -name "*kernel_trace.csv" | xargs ls -t | head -1); python tools/condense_trace.py $f > gpurun_out/r3_config3_default_kernel_trace.csv
cp gpurun_out/pmc/trace.json gpurun_out/r3_bench_config3_default_profiled.json
rm -rf gpurun_out/pmc
cat gpurun_out/r3_c5_rehearse*.json | cut -c1-420
