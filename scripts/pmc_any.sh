#!/bin/bash
# Usage: scripts/pmc_any.sh <tag> "<counters>" <python script> [args...]   (run on the GPU box via gpurun)
# One rocprofv3 --pmc pass (kernel-trace only) over any script; counters averaged per launch and kernel name.
set -u
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; rm -rf "$out" /tmp/pmc_$tag; mkdir -p "$out"
timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python3 "$@" > "$out/run.log" 2>&1
tail -2 "$out/run.log"
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" "$out" <<'PY'
import csv, sys, json, collections
f, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
res = {k: dict({m: v / cnt[k][m] for m, v in c.items()}, launches=max(cnt[k].values())) for k, c in acc.items()}
for k, v in res.items():
    if v.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        v["mfma_pipe_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
json.dump(res, open(f"{out}/pmc.json", "w"), indent=1)
key = "GRBM_GUI_ACTIVE" if any("GRBM_GUI_ACTIVE" in v for v in res.values()) else "SQ_WAVE_CYCLES"
for k in sorted(res, key=lambda k: -res[k].get(key, 0) * res[k]["launches"])[:10]:
    print(k[:90]); print("   ", {m: "%.4g" % v for m, v in res[k].items()})
PY
