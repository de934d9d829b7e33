#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing: per basic block the number of VALU / SALU / LDS /
VMEM / SMEM instructions, with the loop nesting implied by backward branches.  Usage:
    python tools/isa/blocks.py v5.s _ZN4ugvc13fused5_kernelILi3ELi16EEEvNS_6V5ArgsE [--min 20]
A proxy for the VALU-issue cost of a change when no GPU is at hand (the scoring pass is VALU-issue bound)."""
import re
import sys


def classify(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm") or op.startswith("s_setpc") or op.startswith("s_swappc"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    min_n = 20
    if "--min" in sys.argv:
        min_n = int(sys.argv[sys.argv.index("--min") + 1])
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith(kern + ":"):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = []          # (label, line, counts)
    cur = {"label": "entry", "line": start, "n": {}, "br": []}
    label_line = {}
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l.startswith("\t.section") or l.startswith(".Lfunc_end") or l.startswith("\t.end_amdhsa") or l.startswith("\t.p2align\t8") :
            if l.startswith(".Lfunc_end"):
                break
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "line": i, "n": {}, "br": []}
            label_line[m.group(1)] = i
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        c = classify(op)
        cur["n"][c] = cur["n"].get(c, 0) + 1
        if c == "branch":
            t = re.search(r"(\.LBB[0-9_]+)", s)
            if t:
                cur["br"].append(t.group(1))
    blocks.append(cur)
    # backward branches = loops
    idx = {b["label"]: k for k, b in enumerate(blocks)}
    loops = []
    for k, b in enumerate(blocks):
        for t in b["br"]:
            if t in idx and idx[t] <= k:
                loops.append((idx[t], k))
    tot = {}
    for b in blocks:
        for c, n in b["n"].items():
            tot[c] = tot.get(c, 0) + n
    print("kernel totals:", " ".join(f"{c}={n}" for c, n in sorted(tot.items())))
    print("loops (head..tail blocks, static instruction counts inside):")
    for h, t in sorted(set(loops), key=lambda x: (x[0], -x[1])):
        acc = {}
        for b in blocks[h:t + 1]:
            for c, n in b["n"].items():
                acc[c] = acc.get(c, 0) + n
        n_all = sum(acc.values())
        if n_all >= min_n:
            print(f"  {blocks[h]['label']:>12} (line {blocks[h]['line']}) .. {blocks[t]['label']:>12} (line {blocks[t]['line']}): " +
                  " ".join(f"{c}={n}" for c, n in sorted(acc.items())))


if __name__ == "__main__":
    main()
