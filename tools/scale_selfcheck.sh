#!/bin/bash
# The multi-GPU path, proven the moment an N-GPU node appears:  bash tools/scale_selfcheck.sh [max_ranks] [variants]
# For N = 1, 2, 4, 8 (up to the GPUs visible): bench.py over N ranks, one process per GPU, launched exactly as the driver
# launches it; --check-rows -1 makes rank 0 compare its whole shard AND every row of the RCCL-gathered callset with the CPU oracle
# on the unsharded tables.  Asserts per N: rccl_nranks == N (what RCCL itself reports), gather_consistent, every-row parity;
# writes the four bench lines to gpurun_out/scale_selfcheck.jsonl and the strong-scaling table to gpurun_out/scale_selfcheck.txt.
# (The same bookkeeping runs on CPU at world 2 / 3 in tests/test_bench_two_ranks_cpu.py with the oracle as the engine.)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
MAXR=${1:-8}; NVAR=${2:-5000000}
O=gpurun_out; mkdir -p $O; : > $O/scale_selfcheck.jsonl
NGPU=$(python -c "
import ctypes
try:
    h = ctypes.CDLL('libamdhip64.so'); n = ctypes.c_int(0); h.hipGetDeviceCount(ctypes.byref(n)); print(n.value)
except Exception: print(0)")
echo "GPUs visible: $NGPU"
bad=0
for N in 1 2 4 8; do
  [ $N -gt $MAXR ] && break
  if [ $N -gt $NGPU ]; then echo "N=$N: skipped ($NGPU GPUs visible)"; continue; fi
  ARGS="--gpus $N --variants $NVAR --steps 30 --warmup 5 --cpu-sample 0 --no-e2e --check-rows -1"
  if [ $N -eq 1 ]; then timeout 1800 python bench.py $ARGS > $O/scale_$N.out 2> $O/scale_$N.err
  else timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py $ARGS > $O/scale_$N.out 2> $O/scale_$N.err; fi
  rc=$?
  grep '^{' $O/scale_$N.out | tail -1 >> $O/scale_selfcheck.jsonl
  python - $N $rc $O/scale_$N.out <<'PY' || bad=$((bad+1))
import json, sys
N, rc, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lines = [l for l in open(path) if l.startswith("{")]
assert rc == 0 and len(lines) == 1, f"N={N}: rc {rc}, {len(lines)} JSON lines"
d = json.loads(lines[0]); p = d["parity"]; c = d["config"]
assert d["n_gpus"] == N and c["rccl_nranks"] == N, f"N={N}: n_gpus {d['n_gpus']}, RCCL reports {c['rccl_nranks']} ranks"
assert p["oracle_slice_bit_exact"] is True and p["gather_consistent"] is True, f"N={N}: parity {p}"
assert N == 1 or p["gathered_all_rows_bit_exact"] is True, f"N={N}: gathered callset differs from the oracle"
print(f"N={N}: {d['value']:.4g} variants/s, {d['ms_per_step']*1e3:.1f} us per step, kernel {d['roofline']['kernel_ms']*1e3:.1f} us, "
      f"{c['variants_per_gpu']} variants per GPU, every row of the gathered callset == oracle")
PY
done | tee $O/scale_selfcheck.txt
python - <<'PY' | tee -a gpurun_out/scale_selfcheck.txt
import json
rows = [json.loads(l) for l in open("gpurun_out/scale_selfcheck.jsonl") if l.strip()]
if rows:
    base = rows[0]["value"]
    for d in rows:
        print(f"n_gpus {d['n_gpus']}: {d['value']:.4g} variants/s = {d['value']/base:.2f} x the 1-GPU rate")
PY
exit $bad
