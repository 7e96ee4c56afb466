#!/bin/bash
# Usage: scripts/pmc_config.sh <method> [steps]        (run on the GPU box via gpurun; round 6)
# HBM-side traffic of one side configuration's training step (scripts/step_config.py vanilla | cobafa | kplanes): two rocprofv3 passes,
# one counter each (FETCH_SIZE, WRITE_SIZE) with the kernel trace, as MI355X_MICROARCH.md prescribes; bytes = (2 * FETCH_SIZE +
# WRITE_SIZE) * 1024 (gfx950 tallies a 128-B request of a 16 B/lane read as 64 B).  Output: gpurun_out/pmc_<method>/traffic.json =
# per kernel bytes and duration per launch, GB/s, launches per step, and the step's total; copy to profiles/round6_<method>_pmc_traffic.json.
set -u
method=${1:-vanilla}; steps=${2:-6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$method; rm -rf "$out"; mkdir -p "$out"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$ctr
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcc_$ctr -o p -- python3 scripts/step_config.py $method $steps > "$out/run_$ctr.log" 2>&1
  tail -1 "$out/run_$ctr.log"
  cp "$(find /tmp/pmcc_$ctr -name '*counter_collection.csv' | head -1)" "$out/$ctr.csv"
  cp "$(find /tmp/pmcc_$ctr -name '*kernel_trace.csv' | head -1)" "$out/trace_$ctr.csv" 2>/dev/null
done
python3 - "$out" "$method" "$steps" <<'PY'
import csv, sys, json, collections
out, method, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
total_steps = steps + 3                                     # scripts/step_config.py: three warm-up steps
def short(k): return k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ctr = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f"{out}/{c}.csv")):
        if r.get("Counter_Name") != c: continue
        k = short(r["Kernel_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    ctr[c] = {k: (v[0] / v[1], v[1]) for k, v in acc.items()}
dur = collections.defaultdict(lambda: [0.0, 0])
try:
    for r in csv.DictReader(open(f"{out}/trace_FETCH_SIZE.csv")):
        k = short(r["Kernel_Name"]); dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[k][1] += 1
except Exception as e:
    print("no kernel trace:", e)
kern = {}
for k in set(ctr["FETCH_SIZE"]) | set(ctr["WRITE_SIZE"]):
    f, nf = ctr["FETCH_SIZE"].get(k, (0.0, 0)); w, nw = ctr["WRITE_SIZE"].get(k, (0.0, 0))
    n = max(nf, nw)
    b = (2 * f + w) * 1024
    d = dur[k][0] / dur[k][1] if dur[k][1] else None
    kern[k] = {"bytes_per_launch": b, "fetch_kb": f, "write_kb": w, "launches": n, "launches_per_step": n / total_steps,
               "avg_us_profiled": d / 1e3 if d else None, "gbs": b / d if d else None, "frac_of_hbm_peak": b / d / 8000.0 if d else None}
steady = {k: v for k, v in kern.items() if v["launches"] >= total_steps}        # (the occupancy refresh of step 0 and torch's one-off kernels dropped)
per_step = sum(v["bytes_per_launch"] * v["launches"] for v in steady.values()) / total_steps
res = {"method": method, "steps_profiled": total_steps, "bytes_per_step": per_step,
       "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch, separate --pmc passes (MI355X_MICROARCH.md, HBM); steady state: kernels launched every step",
       "command": f"scripts/pmc_config.sh {method} {steps}  (rocprofv3 --pmc <one counter> --kernel-trace -- python3 scripts/step_config.py {method} {steps})",
       "kernels": dict(sorted(steady.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"]))}
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print("bytes per step: %.3f GB" % (per_step / 1e9))
for k, v in list(res["kernels"].items())[:12]:
    print("%8.3f GB/launch x %5.2f per step  %7.1f us  %5.0f GB/s  %s" % (v["bytes_per_launch"] / 1e9, v["launches_per_step"], v["avg_us_profiled"] or 0, v["gbs"] or 0, k[:80]))
PY
