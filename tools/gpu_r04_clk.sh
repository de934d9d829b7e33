#!/bin/bash
# phase clocks of every wave of two workgroups (library prebuilt with -DUGVC_PHASE_CLOCK as tools/ab/clk.so): which waves finish last.
# Several passes run; the LAST pass's line of every (workgroup, wave) is kept (warm caches / TLB).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
cp variantcalling_amd/libugvc_mi355x.so /tmp/keep.so
cp tools/ab/clk.so variantcalling_amd/libugvc_mi355x.so
python bench.py --steps 1 --warmup 4 --spinup 0 --cpu-sample 0 --no-e2e --check-rows 0 $CLK_ARGS 2>&1 | grep -E "^clk|^iclk|^fclk|issue" > /tmp/clk_all.txt
cp /tmp/keep.so variantcalling_amd/libugvc_mi355x.so
python - <<'PY' > gpurun_out/${CLK_OUT:-r04_phase_clocks.txt}
import re
last = {}
for l in open("/tmp/clk_all.txt"):
    m = re.match(r"(\w+) b(\d+) w(\d+)", l)
    key = (m.group(1), int(m.group(2)), int(m.group(3))) if m else ("z", 0, 0)
    last[key] = l.rstrip()
for k in sorted(last, key=lambda k: (k[1], k[0] == "fclk", k[2])):
    print(last[k])
PY
cat gpurun_out/${CLK_OUT:-r04_phase_clocks.txt}
