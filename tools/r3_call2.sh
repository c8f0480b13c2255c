set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/r3_c2_bench_n1.json 2> gpurun_out/r3_c2_bench_n1.err; echo "bench rc=$?"
for g in 8x1 1x8 2x4 4x2; do
  MM_BENCH_REHEARSE_WORLD=8 MM_SHARD_GRID=$g timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c2_rehearse8_$g.json 2> gpurun_out/r3_c2_rehearse8_$g.err; echo "rehearse $g rc=$?"
done
MM_BENCH_REHEARSE_WORLD=8 MM_BENCH_SHARD_LOOKAHEAD=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c2_rehearse8_8x1_lookahead.json 2> gpurun_out/r3_c2_rehearse8_la.err; echo "rc=$?"
MM_BENCH_REHEARSE_WORLD=8 MM_SHARD_GRID=1x8 MM_BENCH_SHARD_LOOKAHEAD=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c2_rehearse8_1x8_lookahead.json 2>/dev/null; echo "rc=$?"
MM_BENCH_REHEARSE_WORLD=2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c2_rehearse2.json 2>/dev/null
MM_BENCH_REHEARSE_WORLD=4 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c2_rehearse4.json 2>/dev/null
MM_BENCH_REHEARSE_WORLD=2 MM_BENCH_SHARD_LOOKAHEAD=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c2_rehearse2_lookahead.json 2>/dev/null
MM_BENCH_REHEARSE_WORLD=4 MM_BENCH_SHARD_LOOKAHEAD=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c2_rehearse4_lookahead.json 2>/dev/null
cat gpurun_out/r3_c2_rehearse*.json | cut -c1-500
