#!/bin/bash
# On the GPU box: the dual form against the single form of k_screen_mx per set size (MM_MX_DUAL_MAX = 17 / 0), separate build tree.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$(mktemp -d /tmp/mm_exp.XXXXXX)
trap 'rm -rf "$W"' EXIT
cp -r "$R/multimoda-rs_amd" "$R/multimoda_rs_amd.py" "$R/include" "$R/oracle" "$R/tools" "$R/tests" "$R/__graft_entry__.py" "$W/"
K="$W/multimoda-rs_amd/csrc/mm_kernels.hip"
cd "$W"
SIZES=${1:-"64 96 128 160 192 208"}
for m in 17 0; do
  sed -i "s/#define MM_MX_DUAL_MAX [0-9]*/#define MM_MX_DUAL_MAX $m/" "$K"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>"$W/build.err" || { echo "build failed"; tail -5 "$W/build.err"; continue; }
  timeout -k 10 200 python tools/bench_mx_sizes.py $SIZES 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('MM_MX_DUAL_MAX=$m:', ' '.join('%d pts: %.1f' % (r['points'], r['matrix_ns_per_tile_per_simd']) for r in d['sizes']), 'ns per tile per SIMD')"
done
