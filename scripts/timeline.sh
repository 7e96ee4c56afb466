#!/bin/bash
# Usage: scripts/timeline.sh <tag> <python script> [args...]  -- kernel trace of a script (GPU box); prints the launches of a ~T_MS window
# between two launches of the marker kernel T_MARK (default sample_pack_kernel = one training step), the T_BACK-th from the end: start offset, duration, queue, kernel
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tl_$tag
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o $tag -- python3 "$@" 2>&1 | grep -v "^W2\|rocprofiler\|^E2" | tail -3
python3 - /tmp/tl_$tag <<'PY'
import csv, glob, sys, os
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
mark = os.environ.get("T_MARK", "sample_pack_kernel")
hits = [k for k, r in enumerate(rows) if mark in r["Kernel_Name"]]
back = int(os.environ.get("T_BACK", "3"))
k0, k1 = hits[-back - 1], hits[-back]
at = int(rows[k0]["Start_Timestamp"])
for r in rows[k0:k1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us +%8.1f  q%-3s %s" % ((s - at) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]))
PY
