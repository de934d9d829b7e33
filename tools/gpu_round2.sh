#!/bin/bash
# One GPU-box visit: GPU test suite, then the headline bench on the production path and on v3.
# usage: tools/gpu_round2.sh [tag]   -> gpurun_out/<tag>.*
tag=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -150 > gpurun_out/$tag.tests.log
tail -5 gpurun_out/$tag.tests.log
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/$tag.bench.json 2> gpurun_out/$tag.bench.err
cat gpurun_out/$tag.bench.json | head -c 1500; echo
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --variant 65536 > gpurun_out/$tag.bench_v3.json 2>> gpurun_out/$tag.bench.err
python - <<'PY'
import json,glob,sys
for f in sorted(glob.glob('gpurun_out/*bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],4),'ms', d['roofline']['kernel_ms'], d['parity'])
    except Exception as e: print(f,'unreadable',e)
PY
tail -3 gpurun_out/$tag.bench.err
