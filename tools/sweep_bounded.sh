#!/bin/bash
# On the GPU box: rebuild with different bounded-screen constants and time bench.py --precision bounded.
# usage: bash tools/sweep_bounded.sh "<kLbCandStep> <kLbRP> <kLbListRP>" ...
set -o pipefail
K=multimoda-rs_amd/csrc/mm_kernels.hip
cp $K /tmp/mm_kernels.orig
for v in "$@"; do
  set -- $v
  cp /tmp/mm_kernels.orig $K
  sed -i "s/static constexpr int kLbCandStep = [0-9]*;/static constexpr int kLbCandStep = $1;/; s/static constexpr int kLbRP = [0-9]*;/static constexpr int kLbRP = $2;/; s/static constexpr int kLbListRP = [0-9]*;/static constexpr int kLbListRP = $3;/" $K
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || { echo "build failed for $v"; continue; }
  timeout -k 10 200 python bench.py --no-cpu-baseline --precision bounded --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['config']['bounded_screen']
print('cand_step $1 RP $2 listRP $3: %.3f ms/step  r1 %.1f%% r2 %.1f%% r3 %.1f%% screened %.2f%%' % (d['ms_per_step'], 100*b['bounded_round1']/b['offered'], 100*b['bounded_round2']/b['offered'], 100*b['bounded_round3']/b['offered'], 100*b['screened']/b['offered']))"
done
cp /tmp/mm_kernels.orig $K
