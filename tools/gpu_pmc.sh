#!/bin/bash
# rocprofv3 passes for the dominant kernel: kernel trace + stats, then PMC counters in their own runs
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --workload ${1:-config3}"
echo "== kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH > $O/trace.json 2> $O/trace.err; echo rc=$?
# (BENCH takes bench.py's default precision: the matrix-pipe screen; append e.g. "--precision fast" through MM_PMC_ARGS)
BENCH="$BENCH ${MM_PMC_ARGS:-}"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  echo "== pmc $c"; timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$tag -- $BENCH > $O/$tag.json 2> $O/$tag.err; rc=$?; echo rc=$rc; [ $rc -ge 124 ] && exit $rc
done
find $O -name "*.csv" | head -30
