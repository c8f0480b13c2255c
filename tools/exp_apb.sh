#!/bin/bash
# On the GPU box: the tiles-per-wave-per-item target of the matrix-pipe screen's work decomposition (mm_engine.cpp, apb_of),
# separate build tree.  usage: bash tools/exp_apb.sh "578 867 1156"
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$(mktemp -d /tmp/mm_exp.XXXXXX)
trap 'rm -rf "$W"' EXIT
cp -r "$R/multimoda-rs_amd" "$R/multimoda_rs_amd.py" "$R/include" "$R/oracle" "$R/tools" "$R/tests" "$R/__graft_entry__.py" "$R/bench.py" "$R/profiles" "$W/"
E="$W/multimoda-rs_amd/csrc/mm_engine.cpp"
cd "$W"
for v in ${1:-578 867 1156}; do
  sed -i "s/std::max<int64_t>(2, ([0-9]* + tiles - 1) \/ tiles)/std::max<int64_t>(2, ($v + tiles - 1) \/ tiles)/; s/int apb = (int)std::min<int64_t>([0-9]*, std::max<int64_t>(1, (A + target_wgs - 1) \/ target_wgs));/int apb = (int)std::min<int64_t>(8, std::max<int64_t>(1, (A + target_wgs - 1) \/ target_wgs));/" "$E"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>"$W/build.err" || { echo "build failed"; tail -5 "$W/build.err"; continue; }
  s=$(timeout -k 10 200 python tools/bench_mx_sizes.py 208 320 521 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' '.join('%d pts: %.1f' % (r['points'], r['matrix_ns_per_tile_per_simd']) for r in d['sizes']))")
  b=$(timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step launch %.3f' % (d['ms_per_step'], d['roofline']['issue']['launch_ms']))")
  echo "tiles per wave per item >= $v: $s | config3 $b"
done
