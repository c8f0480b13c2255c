set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_c_host.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r3_c1_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3_c1_tests.log
tail -5 gpurun_out/r3_c1_tests.log
timeout -k 10 300 python tools/rccl_smoke.py > gpurun_out/r3_c1_rccl_smoke.txt 2>&1; echo "smoke rc=$?"
tail -12 gpurun_out/r3_c1_rccl_smoke.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/r3_c1_bench_n1.json 2> gpurun_out/r3_c1_bench_n1.err; echo "bench rc=$?"
for g in 8x1 1x8 2x4 4x2; do
  MM_BENCH_REHEARSE_WORLD=8 MM_SHARD_GRID=$g timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c1_rehearse8_$g.json 2> gpurun_out/r3_c1_rehearse8_$g.err; echo "rehearse $g rc=$?"
done
MM_BENCH_REHEARSE_WORLD=2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c1_rehearse2.json 2>/dev/null
MM_BENCH_REHEARSE_WORLD=4 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c1_rehearse4.json 2>/dev/null
MM_BENCH_REHEARSE_WORLD=8 MM_EXCHANGE=device timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_c1_rehearse8_torch.json 2>/dev/null
cat gpurun_out/r3_c1_bench_n1.json | cut -c1-400
cat gpurun_out/r3_c1_rehearse*.json | cut -c1-600
ls /sys/class/drm/ ; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -20; rocm-smi --showclocks 2>&1 | head -30
