#!/bin/bash
# On the GPU box: TIMING-ONLY variants of k_screen_mx (wrong results, same instruction stream minus one part) built into a
# separate directory, to see what each part of a candidate costs at a given set size.  usage: bash tools/exp_mx.sh "208 521"
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$(mktemp -d /tmp/mm_exp.XXXXXX)
trap 'rm -rf "$W"' EXIT
cp -r "$R/multimoda-rs_amd" "$R/multimoda_rs_amd.py" "$R/include" "$R/oracle" "$R/tools" "$R/tests" "$R/__graft_entry__.py" "$W/"
K="$W/multimoda-rs_amd/csrc/mm_kernels.hip"
G="$W/tools/gen_screen_mx.py"
cp "$K" "$W/k.orig"; cp "$G" "$W/g.orig"
cd "$W"
run() {
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>"$W/build.err" || { echo "build failed: $1"; tail -5 "$W/build.err"; return; }
  timeout -k 10 200 python tools/bench_mx_sizes.py $SIZES 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1:', ' '.join('%d pts: %.3f ms %.1f ns/tile%s' % (r['points'], r['matrix_ms'], r['matrix_ns_per_tile_per_simd'], '' if r['identical_winners'] else ' (wrong)') for r in d['sizes']))"
}
SIZES=${1:-"208 521"}
run baseline
# E1: the columns are rotated and split for the first candidate of a wave only
cp "$W/k.orig" "$K"
python3 - "$K" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
old="""                    const int j = lane + 64 * q;
                    if (j < NB) {
                        const float bx = __builtin_fmaf(tx[q], c, -(ty[q] * s));"""
assert old in s
s=s.replace(old, old.replace("if (j < NB) {", "if (j < NB && k == wave) {"),1)
open(p,'w').write(s)
PY
run "E1 no per-candidate rotation/split"
cp "$W/k.orig" "$K"
# E3: no blocking reduction of the last row tile
python3 - "$G" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
old='''        s = self.s
        assert self.red is None and t not in self.busy and u not in self.busy
        for v in range(16):'''
assert old in s
s=s.replace(old,'''        s = self.s
        assert self.red is None and t not in self.busy and u not in self.busy
        self.busy.discard(buf)
        return
        for v in range(16):''',1)
open(p,'w').write(s)
PY
python3 "$G" > /dev/null && run "E3 no blocking reduction of the last row tile"
cp "$W/g.orig" "$G"
# E4: no row reduction at all
python3 - "$G" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
old='''            pipe.red = [X[(k - 1) & 1], [], red_off]'''
assert old in s
s=s.replace(old,'''            pipe.busy.discard(X[(k - 1) & 1])''',1)
old2='''        s = self.s
        assert self.red is None and t not in self.busy and u not in self.busy
        for v in range(16):'''
s=s.replace(old2,'''        s = self.s
        assert self.red is None and t not in self.busy and u not in self.busy
        self.busy.discard(buf)
        return
        for v in range(16):''',1)
open(p,'w').write(s)
PY
python3 "$G" > /dev/null && run "E4 no row reduction at all"
cp "$W/g.orig" "$G"; python3 "$G" > /dev/null
