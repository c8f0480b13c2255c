set -x
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/r3_c17_bench_default.json 2> gpurun_out/r3_c17_bench_default.err; echo "bench default rc=$?"; tail -2 gpurun_out/r3_c17_bench_default.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/r3_c17_bench_n1.json 2>/dev/null; echo "rc=$?"
for n in 8 4 2; do
  MM_BENCH_REHEARSE_WORLD=$n timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c17_rehearse$n.json 2> gpurun_out/r3_c17_rehearse$n.err; echo "rehearse $n rc=$?"
done
MM_BENCH_REHEARSE_WORLD=8 MM_SHARD_GRID=1x8 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c17_rehearse8_1x8.json 2>/dev/null
cd /tmp
O=$R/gpurun_out/pmc; mkdir -p $O
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.json 2> $O/trace.err; echo "trace rc=$?"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$tag -- $B > $O/$tag.json 2> $O/$tag.err; rc=$?; echo "pmc $tag rc=$rc"; [ $rc -ge 124 ] && exit $rc
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/r3_config3_matrix_pmc_summary.csv
f=$(find gpurun_out/pmc/trace -name "*kernel_stats.csv" | xargs ls -t | head -1); cp $f gpurun_out/r3_config3_matrix_kernel_stats.csv
f=$(find gpurun_out/pmc/trace -name "*kernel_trace.csv" | xargs ls -t | head -1); python tools/condense_trace.py $f > gpurun_out/r3_config3_matrix_kernel_trace.csv
cp gpurun_out/pmc/trace.json gpurun_out/r3_bench_config3_matrix_profiled.json
rm -rf gpurun_out/pmc
grep "k_screen_mx" gpurun_out/r3_config3_matrix_pmc_summary.csv | grep ",47617024,"
