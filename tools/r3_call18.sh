set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/r3_c18_bench_n1.json 2>/dev/null; echo "rc=$?"
for n in 8 4 2; do
  MM_BENCH_REHEARSE_WORLD=$n timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c18_rehearse$n.json 2> gpurun_out/r3_c18_rehearse$n.err; echo "rehearse $n rc=$?"
done
MM_BENCH_REHEARSE_WORLD=8 MM_BENCH_FINISHERS=1 MM_BENCH_ENGINES=4 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c18_rehearse8_f1.json 2>/dev/null
MM_BENCH_REHEARSE_WORLD=8 MM_BENCH_FINISHERS=3 MM_BENCH_ENGINES=6 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c18_rehearse8_f3.json 2>/dev/null
MM_BENCH_REHEARSE_WORLD=8 timeout -k 10 300 python bench.py --steps 100 --warmup 5 > gpurun_out/r3_c18_rehearse8_k100.json 2>/dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_bench.py tests/test_bench_pipeline.py -x -q 2>&1 | tail -2
