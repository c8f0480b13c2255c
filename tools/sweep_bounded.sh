#!/bin/bash
# On the GPU box: build variants of the library with different bounded-screen constants into a SEPARATE
# directory (the tracked source and the in-tree library are never touched) and time
# bench.py --precision bounded with each.
# usage: bash tools/sweep_bounded.sh "<kLbCandStep> <kLbRP> <kLbListRP>" ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$(mktemp -d /tmp/mm_sweep.XXXXXX)
trap 'rm -rf "$W"' EXIT
cp -r "$R/multimoda-rs_amd" "$R/multimoda_rs_amd.py" "$R/include" "$R/oracle" "$R/bench.py" "$R/__graft_entry__.py" "$W/"
K="$W/multimoda-rs_amd/csrc/mm_kernels.hip"
cp "$K" "$W/mm_kernels.orig"
cd "$W"
for v in "$@"; do
  set -- $v
  cp "$W/mm_kernels.orig" "$K"
  sed -i "s/static constexpr int kLbCandStep = [0-9]*;/static constexpr int kLbCandStep = $1;/; s/static constexpr int kLbRP = [0-9]*;/static constexpr int kLbRP = $2;/; s/static constexpr int kLbListRP = [0-9]*;/static constexpr int kLbListRP = $3;/" "$K"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || { echo "build failed for $v"; continue; }
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --precision bounded --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['config']['bounded_screen']
print('cand_step $1 RP $2 listRP $3: %.3f ms/step  r1 %.1f%% r2 %.1f%% r3 %.1f%% screened %.2f%%' % (d['ms_per_step'], 100*b['bounded_round1']/b['offered'], 100*b['bounded_round2']/b['offered'], 100*b['bounded_round3']/b['offered'], 100*b['screened']/b['offered']))"
done
