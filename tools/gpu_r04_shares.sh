#!/bin/bash
# weighted static tile shares + priority policies: GPU suite, A/B against tools/ab/base_walk6.so, variants, per-wave clocks
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04_shares_pytest.raw 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04_shares_pytest.raw | tail -25 > gpurun_out/r04_shares_pytest.txt
tail -4 gpurun_out/r04_shares_pytest.txt
fi
run() { python bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-e2e "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step %.4f kernel_ms %.4f p5 %.4f p50 %.4f p95 %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_p5'], r['kernel_ms_p50'], r['kernel_ms_p95']), d['parity']['oracle_slice_bit_exact'])"; }
{
echo "# 'other' = tools/ab/base_walk6.so (equal shares, 3 indel waves), 'new' = the working tree (weighted shares, 4 indel-capable waves)"
bash tools/ab/ab.sh tools/ab/base_walk6.so
echo "# --variants 625000"
bash tools/ab/ab.sh tools/ab/base_walk6.so --variants 625000
for n in 5000000 625000; do
for cfg in "UGVC_PRIO=0" "UGVC_PRIO=1" "UGVC_PRIO=2" "UGVC_PRIO=3" "UGVC_WQ=256,256,256,256,256,304" "UGVC_WQ=256,256,256,256,256,256" "UGVC_WQ=288,256,232,232,256,304" "UGVC_WQ=272,256,240,240,288,304" "UGVC_WQ=272,256,240,240,224,304" "UGVC_PRIO=2 UGVC_WQ=256,256,256,256,256,304" "UGVC_PRIO=1 UGVC_WQ=256,256,256,256,256,304"; do
echo -n "n $n $cfg: "; env $cfg bash -c "$(declare -f run); run --variants $n"
done; done
} > gpurun_out/r04_shares_ab.txt 2>&1
cat gpurun_out/r04_shares_ab.txt
WCLK_OUT=r04_wave_clk_shares.txt bash tools/gpu_r04_wclk.sh
