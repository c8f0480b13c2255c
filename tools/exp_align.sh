#!/bin/bash
# On the GPU box: code placement of the generated block (MI355X_MICROARCH.md, "Two waves per SIMD", item 8): every 8-byte
# instruction of the block on an 8-byte boundary (4-byte ones paired with s_nop 0), or all of them 4 bytes off.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$(mktemp -d /tmp/mm_exp.XXXXXX)
trap 'rm -rf "$W"' EXIT
cp -r "$R/multimoda-rs_amd" "$R/multimoda_rs_amd.py" "$R/include" "$R/oracle" "$R/tools" "$R/tests" "$R/__graft_entry__.py" "$R/bench.py" "$R/profiles" "$W/"
G="$W/tools/gen_screen_mx.py"
cp "$G" "$W/g.orig"
cd "$W"
run() {
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>"$W/build.err" || { echo "build failed: $1"; tail -5 "$W/build.err"; return; }
  s=$(timeout -k 10 200 python tools/bench_mx_sizes.py 208 521 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' '.join('%d pts: %.1f%s' % (r['points'], r['matrix_ns_per_tile_per_simd'], '' if r['identical_winners'] else ' (wrong)') for r in d['sizes']))")
  b=$(timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step launch %.3f' % (d['ms_per_step'], d['roofline']['issue']['launch_ms']))")
  echo "$1: $s | config3 $b"
}
run "as shipped"
for phase in 0 4; do
  cp "$W/g.orig" "$G"
  python3 - "$G" $phase <<'PY'
import sys
p, phase = sys.argv[1], int(sys.argv[2])
s = open(p).read()
# post-process every generated block: 4-byte instructions in pairs, the block on an 8-byte boundary (+ phase)
hook = '''
def _place(out, phase):
    res = [".p2align 3"] + (["s_nop 0"] if phase else [])
    for ln in out:
        res.append(ln)
        op = ln.split()[0]
        if op.startswith("s_") and not ln.rstrip().endswith(":"):
            res.append("s_nop 0")
    return res

'''
s = s.replace("def main():", hook + "def main():", 1)
s = s.replace("                out, regs = generate(nct, carry)\n", "                out, regs = generate(nct, carry)\n                out = _place(out, %d)\n" % phase, 1)
open(p, 'w').write(s)
PY
  python3 "$G" > /dev/null && run "8-byte instructions at offset $phase mod 8"
done
