#!/bin/bash
# The host codec's tests on an AddressSanitizer + UBSan build of libugvc_vcf.so (CPU only): tools/codec_asan.sh [pytest args]
cd "$(dirname "$0")/.."
make -C variantcalling_amd/csrc_host asan > /dev/null || exit 1
ASAN=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
UGVC_VCF_LIB=$PWD/variantcalling_amd/libugvc_vcf_asan.so \
  python -m pytest tests/test_vcf_native.py tests/test_htslib_bgzf.py tests/test_io_host.py -x -q -p no:cacheprovider "$@"
