#!/bin/bash
# On the GPU box: TIMING-ONLY variants of the DUAL form of k_screen_mx (wrong results).  usage: bash tools/exp_mx3.sh "208 521"
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$(mktemp -d /tmp/mm_exp.XXXXXX)
trap 'rm -rf "$W"' EXIT
cp -r "$R/multimoda-rs_amd" "$R/multimoda_rs_amd.py" "$R/include" "$R/oracle" "$R/tools" "$R/tests" "$R/__graft_entry__.py" "$W/"
G="$W/tools/gen_screen_mx.py"
K="$W/multimoda-rs_amd/csrc/mm_kernels.hip"
cp "$G" "$W/g.orig"
sed -i 's/#define MM_MX_DUAL_MAX [0-9]*/#define MM_MX_DUAL_MAX 17/' "$K"
cd "$W"
SIZES=${1:-"208 521"}
run() {
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>"$W/build.err" || { echo "build failed: $1"; tail -5 "$W/build.err"; return; }
  timeout -k 10 200 python tools/bench_mx_sizes.py $SIZES 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1:', ' '.join('%d pts: %.3f ms %.1f ns/tile%s' % (r['points'], r['matrix_ms'], r['matrix_ns_per_tile_per_simd'], '' if r['identical_winners'] else ' (wrong)') for r in d['sizes']))"
}
run "dual baseline"
python3 - "$G" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
old='''        s.ins(f"v_mfma_f32_32x32x16_f16 v[{dt}:{dt + 15}], a[{4 * t}:{4 * t + 3}], v[{a_reg}:{a_reg + 3}], 0",
              reads=rng(a_reg, 4) + rng(AG + 4 * t, 4), writes=rng(dt, 16), mfma=True)'''
assert old in s
s=s.replace(old,"        s.used.update(rng(dt, 16))",1)
open(p,'w').write(s)
PY
python3 "$G" > /dev/null && run "X1 no transposed MFMA (its minima fold stale registers)"
cp "$W/g.orig" "$G"
python3 - "$G" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
old='''            if prev[3]:
                self.finish_row_tile(prev[2])'''
assert old in s
s=s.replace(old,"            pass",1)
open(p,'w').write(s)
PY
python3 "$G" > /dev/null && run "X2 no row-tile finish"
cp "$W/g.orig" "$G"
python3 - "$G" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
old='''    def fold(self, which):'''
new='''    def fold(self, which):
        if which == 1:
            return
        return self.fold_(which)

    def fold_(self, which):'''
assert old in s
s=s.replace(old,new,1)
open(p,'w').write(s)
PY
python3 "$G" > /dev/null && run "X3 no row minima (transposed tile computed, not folded)"
