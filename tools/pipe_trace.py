"""Reads the UGVC_PIPE_TRACE=1 lines of ugvc_filter_variants (csrc/pipeline.hip) from stdin / a file and prints the LAST
call's merged host + device timeline (ms since entry) and the per-phase durations per chunk."""
import sys

L = [l.split() for l in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin) if l.startswith("[pipe]")]
calls = []
for l in L:
    if l[1] == "host" and l[4] == "setup":
        calls.append([])
    if calls:
        calls[-1].append(l)
for c in calls:
    h = [(int(l[3]), l[4], float(l[5])) for l in c if l[1] == "host"]
    print("call: done at %.3f ms" % h[-1][2])
c = calls[-1]
host = [(int(l[3]), l[4], float(l[5])) for l in c if l[1] == "host"]
dev = [(int(l[3]), l[4], float(l[5])) for l in c if l[1] == "dev"]
t0 = host[0][2]
ev = [(t, "H", k, w) for k, w, t in host] + [(t + t0, "D", k, w) for k, w, t in dev]
brief = "--brief" in sys.argv
if not brief:
    for t, kind, k, w in sorted(ev):
        print(f"{t:8.3f} {kind} {k:3d} {w}")
# per-phase host durations
d = {}
prev = None
for k, w, t in host:
    if prev is not None:
        d.setdefault(prev[1] + "->" + w, []).append(t - prev[2])
    prev = (k, w, t)
for key, v in d.items():
    v = sorted(v)
    print(f"{key:28s} n={len(v):3d} median {v[len(v)//2]:.3f} sum {sum(v):.3f}")
