#!/usr/bin/env python3
"""Build-time check of the register-indexed predicates of csrc/kernels_gemm.hip (UGVC_PRED_ASM): the compare reads the row's
feature vector through VGPR index mode relative to `x0`, which must therefore be the FIRST register of the 32-register tuple `xv`.
Every instance leaves `; ugvc_pred x0=vN xv=v[A:B]` in the listing; this script fails the build unless N == A and B == A + 31 in
all of them (a compiler update that moves the vector shows up HERE, not as wrong margins).  Usage: check_pred_asm.py listing.s"""
import re
import sys

text = open(sys.argv[1]).read()
hits = re.findall(r";\s*ugvc_pred x0=v(\d+) xv=v\[(\d+):(\d+)\]", text)
odd = [l for l in re.findall(r";\s*ugvc_pred[^\n]*", text) if not re.match(r";\s*ugvc_pred x0=v\d+ xv=v\[\d+:\d+\]\s*$", l)]
bad = [(x0, a, b) for x0, a, b in hits if int(x0) != int(a) or int(b) != int(a) + 31]
if not hits or bad or odd:
    print(f"check_pred_asm: {len(hits)} instances, {len(bad)} with x0 off the tuple's first register {bad[:5]}, {len(odd)} unparsed {odd[:3]}", file=sys.stderr)
    sys.exit(1)
print(f"check_pred_asm: {len(hits)} predicate instances, x0 is xv's first register in all of them")
