#!/bin/bash
# Usage: scripts/pmc_kernel.sh <tag> "<counters>" <kernel-substring> -- python3 script.py ...
set -u
tag=$1; ctrs=$2; pat=$3; shift 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pk_$tag
timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pk_$tag -o p -- "$@" > /tmp/pk_$tag.log 2>&1
f=$(find /tmp/pk_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
f, pat = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print("   %-28s %16.0f (last of %d)" % (c, v[-1], len(v)))
PY
