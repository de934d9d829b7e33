#!/bin/bash
# quick correctness (oracle diff on two callsets) + kernel trace of the bench: tools/gpu_quick.sh <tag> [bench args]
tag=$1; shift
export TMPDIR=/tmp
cd /root/repo
for n in 3000 40000; do timeout 300 python tools/dbg_v5.py 0 $n 2>&1 | grep -E "differ|Error|error|fault" | cut -c 1-200; done
bash tools/gpu_prof.sh $tag "$@"
