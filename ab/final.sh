#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/r3c_bench_default.json 2> gpurun_out/r3c_bench_default.err || { tail -5 gpurun_out/r3c_bench_default.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r3c_bench_default.json'));print('default',d['ms_per_step'],d['value'],d['roofline']['dominant_launch']['avg_ms'],d['fast_screen']['identical_to_headline_result'])"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r3c_reh_1.json 2> gpurun_out/r3c_reh_1.err || exit 1
for n in 2 4 8; do
  MM_BENCH_REHEARSE_WORLD=$n timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r3c_reh_$n.json 2> gpurun_out/r3c_reh_$n.err || { tail -5 gpurun_out/r3c_reh_$n.err; exit 1; }
done
python - <<'PY'
import json
for n in (1,2,4,8):
    d=json.load(open(f"gpurun_out/r3c_reh_{n}.json"))
    r=d.get("rehearsal_rank0") or d
    print(n, d.get("ms_per_step"), json.dumps(r)[:300] if n>1 else d["roofline"]["dominant_launch"]["avg_ms"])
PY
