set -x
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3_tl
mkdir -p $O
export MM_BENCH_REHEARSE_WORLD=8
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/nola -- python3 $R/bench.py --steps 12 --warmup 4 > $O/nola.json 2> $O/nola.err; echo rc=$?
MM_BENCH_SHARD_LOOKAHEAD=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/la -- python3 $R/bench.py --steps 12 --warmup 4 > $O/la.json 2> $O/la.err; echo rc=$?
cd $R
for t in nola la; do
  f=$(find $O/$t -name "*kernel_trace.csv" | head -1)
  python tools/trace_timeline.py $f > gpurun_out/r3_c3_timeline_$t.csv
  wc -l gpurun_out/r3_c3_timeline_$t.csv
done
cat $O/nola.json $O/la.json | cut -c1-400
rm -rf $O/nola $O/la
