#!/bin/bash
# GPU-box suite runner (gpurun):  tools/gpu_suite.sh <tag> <mode> [<mode> ...]
#   fresh    the driver's sequence on a fresh lease: smoke() as the box's FIRST GPU process, then `pytest -m gpu -x`
#   guard1   the whole -m gpu suite, one process per test file, every device buffer's END on an unmapped 2 MiB range
#   guard2   ... every buffer's START behind an unmapped range
#   poison   ... every new buffer filled with 0xA5 (no guard mapping)
#   sync     ... plain allocator, every launch named and waited for
#   ldsrand  the small GPU tests with other kinds of LDS garbage in front of every launch
#   fuzzmore the randomised differential tests over three more sets of draws, under guard1 + poison
#   quick    smoke first, then the canary and the small GPU tests (one lease's short form of `fresh`)
# Logs: gpurun_out/<tag>_<mode>.txt (+ a one-line verdict per mode in gpurun_out/<tag>_summary.txt)
set -u
tag=$1; shift
out=gpurun_out; mkdir -p $out
sum=$out/${tag}_summary.txt
echo "== $tag $(date -u +%FT%TZ) $(git rev-parse --short HEAD 2>/dev/null) lib sha $(sha256sum variantcalling_amd/libugvc_mi355x.so | cut -c1-16)" >> $sum
per_file() {   # per_file <log> <env...>
    local log=$1; shift
    local bad=0 total=0
    for f in tests/test_*.py; do
        grep -q "mark.gpu" $f || continue
        echo "---- $f" >> $log
        env "$@" timeout 420 python -m pytest $f -q -s -m gpu -p no:cacheprovider >> $log 2>&1
        rc=$?
        line=$(grep -E "passed|failed|no tests ran|Memory access fault" $log | tail -1 | cut -c1-160)
        echo "rc=$rc $f :: $line" >> $sum
        [ $rc -ne 0 ] && [ $rc -ne 5 ] && bad=$((bad+1))
        total=$((total+1))
    done
    echo "$bad of $total files not green ($*)" >> $sum
}
for mode in "$@"; do
    log=$out/${tag}_${mode}.txt; : > $log
    case $mode in
    fresh)   # FRESH_ORDER=driver: pytest first, then smoke (the driver's order); default: smoke as the lease's first GPU process
        if [ "${FRESH_ORDER:-smoke}" = driver ]; then
            timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider >> $log 2>&1
            echo "fresh pytest (first GPU process) rc=$? :: $(grep -E 'passed|failed' $log | tail -1)" >> $sum
            timeout 600 python -c 'import __graft_entry__ as e; e.smoke(); print("__SMOKE_OK__")' >> $log 2>&1
            echo "fresh smoke rc=$?" >> $sum
        else
            timeout 600 python -c 'import __graft_entry__ as e; e.smoke(); print("__SMOKE_OK__")' >> $log 2>&1
            echo "fresh smoke (first GPU process) rc=$?" >> $sum
            timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider >> $log 2>&1
            echo "fresh pytest rc=$? :: $(grep -E 'passed|failed' $log | tail -1)" >> $sum
        fi ;;
    quick)   # one lease's short form of the driver's sequence: smoke FIRST, then the canary and the small GPU tests
        timeout 600 python -c 'import __graft_entry__ as e; e.smoke(); print("__SMOKE_OK__")' >> $log 2>&1
        echo "quick smoke (first GPU process of the lease) rc=$?" >> $sum
        timeout 900 python -m pytest tests/test_00_gpu_canary.py tests/test_annotate.py tests/test_gpu_fuzz.py tests/test_gpu_sec.py tests/test_gpu_eval.py tests/test_gpu_pipelines.py -x -q -m gpu -p no:cacheprovider >> $log 2>&1
        echo "quick pytest rc=$? :: $(grep -E 'passed|failed' $log | tail -1)" >> $sum ;;
    ldsrand) # the small GPU tests with other kinds of LDS garbage in front of every launch: zeros, ones, small and wide random words
        for pat in 00000000 00000001 r2 r8 r32; do
            echo "---- UGVC_POISON_LDS=$pat" >> $log
            UGVC_POISON=1 UGVC_POISON_LDS=$pat timeout 900 python -m pytest tests/test_00_gpu_canary.py tests/test_annotate.py tests/test_gpu_fuzz.py tests/test_gpu_sec.py tests/test_gpu_eval.py tests/test_gpu_pipelines.py -q -s -m gpu -p no:cacheprovider >> $log 2>&1
            echo "ldsrand $pat rc=$? :: $(grep -E 'passed|failed|Memory access fault' $log | tail -1 | cut -c1-160)" >> $sum
        done ;;
    fuzzmore) # the randomised differential tests over other draws, under the guard-end + poison mode
        for k in 1 2 3; do
            echo "---- UGVC_FUZZ_OFFSET=$k" >> $log
            UGVC_FUZZ_OFFSET=$k UGVC_GUARD=1 UGVC_POISON=1 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -s -m gpu -p no:cacheprovider >> $log 2>&1
            echo "fuzzmore offset $k (guard1 + poison) rc=$? :: $(grep -E 'passed|failed|Memory access fault' $log | tail -1 | cut -c1-160)" >> $sum
        done ;;
    guard1) per_file $log UGVC_GUARD=1 UGVC_POISON=1 UGVC_DEBUG_SYNC=1 ;;
    guard2) per_file $log UGVC_GUARD=2 UGVC_POISON=2 UGVC_DEBUG_SYNC=1 ;;
    poison) per_file $log UGVC_POISON=1 ;;
    sync)   per_file $log UGVC_DEBUG_SYNC=1 ;;
    esac
done
cat $sum | tail -60
