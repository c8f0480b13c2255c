#!/bin/bash
# rocprofv3 --pmc passes (each in its own run, kernel trace only) for an arbitrary bench script:
#   bash tools/gpu_pmc_cmd.sh <tag> <python script> [args...]     -> gpurun_out/pmc_<tag>/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  echo "== pmc $c"; timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$tag -- python3 $R/"$@" > $O/$tag.json 2> $O/$tag.err; rc=$?; echo rc=$rc; [ $rc -ge 124 ] && exit $rc
done
exit 0
