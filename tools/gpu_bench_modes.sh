#!/bin/bash
# every bench workload once + the new GPU tests: tools/gpu_bench_modes.sh <tag>
tag=${1:-m1}
export TMPDIR=/tmp
cd /root/repo; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sliced_context" 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/$tag.filter.json 2> gpurun_out/$tag.filter.err; tail -c 1800 gpurun_out/$tag.filter.json; echo
for w in c2 pileup sec_apply c5_gemm; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/$tag.$w.json 2> gpurun_out/$tag.$w.err
  python - "$tag" "$w" <<'PY'
import json, sys
tag, w = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/{tag}.{w}.json").read().strip().splitlines()[-1])
    print(w, "value", f"{d['value']:.4g}", d["unit"], "ms", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), d.get("parity"))
except Exception as e:
    print(w, "FAILED", e); print(open(f"gpurun_out/{tag}.{w}.err").read()[-800:])
PY
done
