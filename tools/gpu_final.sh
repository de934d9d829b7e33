#!/bin/bash
# round-end validation on the GPU box: full GPU test suite, smoke(), the default bench line, the other bench
# workloads, then the rocprofv3 trace + FETCH_SIZE / WRITE_SIZE passes.  usage: gpurun -- 'bash tools/gpu_final.sh <tag>'
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-r02}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1
grep -E "passed|failed|error" gpurun_out/${TAG}_pytest_gpu.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py ) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -3 gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench.json | cut -c 1-400
for w in c2 pileup sec_apply c5_gemm; do
  python bench.py --workload $w --cpu-sample 0 > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err
  tail -1 gpurun_out/${TAG}_bench_$w.json | cut -c 1-240
done
bash tools/gpu_profile.sh $TAG
python tools/make_traffic_json.py gpurun_out/prof_$TAG gpurun_out/hbm_traffic_$TAG.json
