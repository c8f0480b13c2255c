#!/bin/bash
# tests on the new kernel, then same-box A/B of the two libraries
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_property.py -m gpu -x -q -k "matrix or mx or property or precision" > gpurun_out/r3b_tests.txt 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r3b_tests.txt; tail -3 gpurun_out/r3b_tests.txt
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  for v in new v3; do
    MM_LIB_PATH=$GRAFT_REPO_ROOT/ab/$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/r3b_ab_${v}_$i.json 2> gpurun_out/r3b_ab_${v}_$i.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/r3b_ab_${v}_$i.json"))
r=d["roofline"]
print("$v $i", round(d["ms_per_step"],3), round(d["value"]/1e6,2), r["dominant_launch"]["avg_ms"], d.get("fast_screen",{}).get("identical_to_headline_result"))
PY
  done
done
