#!/bin/bash
# round 6, lease A: the driver's order on a fresh lease (smoke first, then the -m gpu suite), the default bench line with
# `other_workloads`, the CLI on 5 M records under both copy paths
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth; O=gpurun_out; mkdir -p $O
nolog() { grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"; }
{ echo "== smoke (first GPU process of the lease)"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | nolog | tail -6
  echo "== pytest -m gpu -x -q"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | nolog | tail -6; } > $O/r06_lease_a_suite.txt 2>&1
tail -3 $O/r06_lease_a_suite.txt
python bench.py > $O/r06_bench.json 2> $O/r06_bench.err; tail -c 3000 $O/r06_bench.json
{ for rep in 1 2; do for mode in slots direct; do
    echo "== UGVC_COPY=$mode rep $rep"; UGVC_COPY=$mode python tools/bench_pipeline.py 5000000 2>&1 | grep -v "^\[vcf\]" | nolog | tail -8
  done; done; } > $O/r06_cli_copy_ab.txt 2>&1
cat $O/r06_cli_copy_ab.txt
