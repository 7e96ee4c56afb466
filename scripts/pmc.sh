#!/bin/bash
# Usage: scripts/pmc.sh <tag> [bench args...]   (run on the GPU box via gpurun)
# HBM traffic per kernel from the L2 memory-side counters, one counter per pass (MI355X_MICROARCH.md, HBM):
#   bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   -- FETCH_SIZE on gfx950 counts 64 B per 128-B request.
set -u
tag=${1:-pmc}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; rm -rf "$out"; mkdir -p "$out"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o p -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_$ctr.log" 2>&1
  f=$(find /tmp/pmc_$ctr -name "*counter_collection.csv" | head -1)
  ls -la /tmp/pmc_$ctr/* | head -5
  python3 - "$f" "$ctr" "$out" <<'PY'
import csv, sys, json, collections
f, ctr, out = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != ctr: continue
    k = r["Kernel_Name"]
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
res = {k: {"avg": v[0] / v[1], "dispatches": v[1]} for k, v in acc.items()}
json.dump(res, open(f"{out}/{ctr}.json", "w"), indent=1)
top = sorted(res.items(), key=lambda kv: -kv[1]["avg"] * kv[1]["dispatches"])[:12]
for k, v in top: print(ctr, "%12.1f KB avg x %4d  %s" % (v["avg"], v["dispatches"], k[:90]))
PY
done
