#!/bin/bash
# Usage: scripts/profile.sh <tag> [bench args...]   (run on the GPU box via gpurun)
# Kernel-trace + stats of bench.py; keeps only the small CSV summaries under gpurun_out/<tag>/.
set -u
tag=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf /tmp/prof_$tag "$out"; mkdir -p "$out"
# TN_SIDE_PLAN=0: the next step's sampler pass in line with the step instead of on its own stream -- kernel durations in the stats file
# then never overlap (side by side, batch_plan_scan_kernel shows the time it queues behind the chain kernel's waves, and
# summing TotalDurationNs over-counts the step by ~0.6 ms)
TN_SIDE_PLAN=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
find /tmp/prof_$tag -name "*stats*.csv" -exec cp {} "$out/" \;
ls -la /tmp/prof_$tag/* | head; grep '^{' "$out/bench.log" | tail -1 > "$out/bench.json"
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6)
    for r in rows[:28]:
        print("%9.3f ms %6s calls %9.1f us avg %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:110]))
PY
cat "$out/bench.json" | head -c 1500
