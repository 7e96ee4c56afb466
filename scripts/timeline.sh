#!/bin/bash
# Usage: scripts/timeline.sh <tag> <python script> [args...]  -- kernel trace of a script (GPU box); prints the launches of a ~T_MS window
# (default 18 ms) that starts T_BACK ms (default 50) before the last launch ends: start offset, duration, queue, kernel
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tl_$tag
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o $tag -- python3 "$@" 2>&1 | grep -v "^W2\|rocprofiler\|^E2" | tail -3
python3 - /tmp/tl_$tag <<'PY'
import csv, glob, sys, os
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
at = t1 - float(os.environ.get("T_BACK", "50")) * 1e6
win = float(os.environ.get("T_MS", "18")) * 1e6
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < at or s > at + win: continue
    print("%9.1f us +%8.1f  q%-3s %s" % ((s - at) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]))
PY
