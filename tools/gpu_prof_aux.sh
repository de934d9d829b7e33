#!/bin/bash
# rocprofv3 kernel trace of the other bench workloads (pileup, sec_apply, c2, c5_gemm) -> gpurun_out/r02_kernel_stats_aux.txt
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
echo "# bash tools/gpu_prof_aux.sh : rocprofv3 --kernel-trace --stats of \`python bench.py --workload <w> --cpu-sample 0\` (one MI355X); name | calls | average us" > gpurun_out/r02_kernel_stats_aux.txt
for w in pileup sec_apply c2 c5_gemm; do
  rm -rf gpurun_out/pa
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pa -o prof -- python bench.py --workload $w --cpu-sample 0 > gpurun_out/pa.log 2>&1
  python - "$w" <<'PY' >> gpurun_out/r02_kernel_stats_aux.txt
import csv, glob, sys
print("==", sys.argv[1])
for f in glob.glob("gpurun_out/pa/**/*kernel_stats.csv", recursive=True):
    for k, r in enumerate(csv.DictReader(open(f))):
        if k < 5: print(f"{r['Name'][:70]:70s} | {r['Calls']:>5s} | {float(r['AverageNs'])/1e3:10.1f}")
PY
done
rm -rf gpurun_out/pa
cat gpurun_out/r02_kernel_stats_aux.txt
