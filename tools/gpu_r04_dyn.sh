#!/bin/bash
# dynamic tile scheduling: GPU suite, A/B against the previous commit's library, the scheduler's variants, per-wave clocks
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04_dyn_pytest.raw 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04_dyn_pytest.raw | tail -25 > gpurun_out/r04_dyn_pytest.txt
tail -4 gpurun_out/r04_dyn_pytest.txt
fi
run() { python bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-e2e "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step %.4f kernel_ms %.4f p5 %.4f p50 %.4f p95 %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_p5'], r['kernel_ms_p50'], r['kernel_ms_p95']), d['parity'])"; }
{
echo "# 'other' = tools/ab/base_walk6.so (static tile shares), 'new' = the working tree (tiles handed out in chunks + stealing)"
bash tools/ab/ab.sh tools/ab/base_walk6.so
echo "# --variants 625000"
bash tools/ab/ab.sh tools/ab/base_walk6.so --variants 625000
for n in 5000000 625000; do
for var in 0 1048576 2097152 3145728; do
echo -n "n $n variant $var (1048576 no stealing, 2097152 one static-size chunk per wave): "; run --variants $n --variant $var
done; done
} > gpurun_out/r04_dyn_ab.txt 2>&1
cat gpurun_out/r04_dyn_ab.txt
WCLK_OUT=r04_wave_clk_dyn.txt bash tools/gpu_r04_wclk.sh
