set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_ccta.py tests/test_gpu_parity.py tests/test_gpu_c_host.py -x -q -m gpu > gpurun_out/r3_c4_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3_c4_tests.log
tail -15 gpurun_out/r3_c4_tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/r3_c4_bench_n1.json 2> gpurun_out/r3_c4_bench_n1.err; echo "bench rc=$?"
for n in 8 4 2; do
  MM_BENCH_REHEARSE_WORLD=$n timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c4_rehearse$n.json 2> gpurun_out/r3_c4_rehearse$n.err; echo "rehearse $n rc=$?"
done
MM_BENCH_REHEARSE_WORLD=8 MM_SHARD_GRID=1x8 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c4_rehearse8_1x8.json 2> gpurun_out/r3_c4_rehearse8_1x8.err
MM_BENCH_REHEARSE_WORLD=8 MM_SHARD_GRID=4x2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c4_rehearse8_4x2.json 2>/dev/null
MM_BENCH_REHEARSE_WORLD=8 MM_BENCH_NO_LOOKAHEAD=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c4_rehearse8_nola.json 2>/dev/null
MM_BENCH_REHEARSE_WORLD=8 timeout -k 10 300 python bench.py --steps 100 --warmup 5 > gpurun_out/r3_c4_rehearse8_k100.json 2>/dev/null
cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_tl; mkdir -p $O
MM_BENCH_REHEARSE_WORLD=8 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/ch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 > $O/ch.json 2> $O/ch.err; echo rc=$?
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py $(find $O/ch -name "*kernel_trace.csv" | head -1) > gpurun_out/r3_c4_timeline_chained.csv
rm -rf $O/ch
cat gpurun_out/r3_c4_rehearse*.json | cut -c1-420
