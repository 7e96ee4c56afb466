#!/bin/bash
# Usage: scripts/pmc_mfma.sh <tag> [bench args...]   (run on the GPU box via gpurun)
# Matrix-pipe occupancy per kernel: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE in one
# counter pass (kernel-trace only, as gpurun requires), summed per kernel name.
set -u
tag=${1:-mfma}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; rm -rf "$out" /tmp/pmc_$tag; mkdir -p "$out"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" "$out" <<'PY'
import csv, sys, json, collections
f, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
res = {}
for k, c in acc.items():
    n = max(cnt[k], 1)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    res[k] = {"launches": n, **{m: v / n for m, v in c.items()},
              # SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over the chip's 1024 SIMDs (check: wgrad4 = 1.1e7 MFMAs x 64 cycles);
              # GRBM_GUI_ACTIVE: cycles, summed over the 8 XCDs -> fraction of the launch the matrix pipes were busy:
              "mfma_pipe_busy_frac": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0)) if gui else None}
json.dump(res, open(f"{out}/mfma.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"])[:14]:
    print("%-70s x%4d  gui %12.0f  mfma_busy %14.0f  busy_cu %14.0f  frac %s" % (k[:70], v["launches"], v.get("GRBM_GUI_ACTIVE", 0),
          v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("SQ_BUSY_CU_CYCLES", 0), v["mfma_pipe_busy_frac"]))
PY
