#!/bin/bash
# Usage: scripts/pmc_sq.sh <tag> "<counters>" [bench args...]   (run on the GPU box via gpurun)
# Any set of (<= 8 SQ) counters per kernel, averaged per launch.
set -u
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; rm -rf "$out" /tmp/pmc_$tag; mkdir -p "$out"
timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python3 bench.py --no-cpu-baseline --no-stages --windows 1 "$@" > "$out/bench.log" 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" "$out" <<'PY'
import csv, sys, json, collections
f, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
res = {k: {m: v / cnt[k][m] for m, v in c.items()} for k, c in acc.items()}
json.dump(res, open(f"{out}/sq.json", "w"), indent=1)
for k in sorted(res, key=lambda k: -res[k].get("SQ_WAVE_CYCLES", 0))[:8]:
    print(k[:80]); print("   ", {m: "%.3g" % v for m, v in res[k].items()})
PY
