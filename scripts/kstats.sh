#!/bin/bash
# Usage: scripts/kstats.sh <tag> <python script> [args...]  -- kernel-trace stats of any script (GPU box)
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/ks_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o $tag -- python3 "$@" 2>&1 | grep -v "^W2\|rocprofiler" | tail -3
python3 - /tmp/ks_$tag <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:int(__import__("os").environ.get("KS_TOP", "12"))]:
        print("%9.3f ms %5s calls %9.1f us avg %9.1f min  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Name"].replace("(anonymous namespace)::", "")[:100]))
PY
