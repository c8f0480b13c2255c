# kernel trace of the bounded search (bench.py --precision bounded) for the variants given: 0 = packed-FMA kernels,
# 11 / 12 / 21 / 22 = k_bound_mx<query tiles per side, candidates per wave>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_bounded_trace
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in ${@:-21 12 22 11 0}; do
  export MM_BENCH_BOUND_MATRIX=$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$m -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --precision bounded > $O/m$m.json 2> $O/m$m.err; echo "variant $m rc=$?"
done
