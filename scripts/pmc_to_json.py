#!/usr/bin/env python3
"""gpurun_out/<tag>/{FETCH_SIZE,WRITE_SIZE}.json (scripts/pmc.sh) -> profiles/<name>.json: HBM-side bytes per launch, per
timed C-ABI entry point (the tags of bench.py's KERNEL_MODEL) and per training step.

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies a 128-B request of a wide (16 B/lane) read as 64 B
(MI355X_MICROARCH.md, HBM section); every streaming read of these kernels is 16 B/lane.  WRITE_SIZE counts 4 B per lane of a
memory-side fp32 atomic (calibrated on scripts/microbench/atomic_patterns.hip), so for the kernel that carries the plane
scatter  lane-atomics = (WRITE_SIZE - plain row stores) / 4 B, the plain stores being the workspace rows it writes
(algorithmic: 260 + 68 rows of 128 B per 32-sample tile).

usage: pmc_to_json.py <gpurun_out/tag> <profiles/name.json> <samples per step>"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
samples = float(sys.argv[3]) if len(sys.argv) > 3 else 1028222.0
f = json.load(open(f"{src}/FETCH_SIZE.json"))
w = json.load(open(f"{src}/WRITE_SIZE.json"))


def short(k):
    return k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


per = {}
for k in set(f) | set(w):
    fk, wk = f.get(k, {"avg": 0.0, "dispatches": 0}), w.get(k, {"avg": 0.0, "dispatches": 0})
    per[short(k)] = {"FETCH_SIZE_KB": fk["avg"], "WRITE_SIZE_KB": wk["avg"], "launches_seen": max(fk["dispatches"], wk["dispatches"]),
                     "bytes": (2 * fk["avg"] + wk["avg"]) * 1024}


def total(*subs):
    return sum(v["bytes"] for k, v in per.items() if any(k.startswith(s) for s in subs))


def find(sub):
    return next((v for k, v in per.items() if k.startswith(sub)), None)


chain = find("mlp_chain_kernel<64, 4, 8, true, false, true, true")
steps = chain["launches_seen"] if chain else 1
entry = {
    "tn_kplanes_mlp_fwd_pair": total("mlp_fwd_kernel<64, true, 12, true, true, true, true", "mlp_fwd_kernel<64, true, 8, true, true, true, true"),
    "tn_kplanes_mlp_bwd_pair:chain": total("mlp_chain_kernel<64, 4, 8, true, false, true, true"),
    "tn_kplanes_mlp_bwd_pair:wgrad": total("mlp_wgrad4_kernel<4", "mlp_wgrad_kernel<64, 1", "wgrad_first_kernel", "wgrad_rc_kernel"),
    "tn_adam_reg_multi": total("adam_reg_multi_kernel"),
    "tn_mlp_bwd_pair": total("mlp_chain_kernel<64, 4, 8, true, false, true, false", "mlp_wgrad4_kernel<4", "mlp_wgrad_kernel<64, 1"),
    "tn_mlp_fwd_stash_pair": total("mlp_fwd_kernel<64, true, 16, true, true, true, false", "mlp_fwd_kernel<64, true, 12, true, true, true, false"),
    "tn_kplanes_bwd": total("kplanes_bwd_kernel"), "tn_kplanes_fwd": total("kplanes_fwd_kernel"),
}
entry = {k: v for k, v in entry.items() if v}
lanes = {}
if chain:
    plain = (samples / 32.0) * (260 + 68) * 128.0
    lanes["tn_kplanes_mlp_bwd_pair:chain"] = max(0.0, chain["WRITE_SIZE_KB"] * 1024 - plain) / 4.0
kb = find("kplanes_bwd_kernel")
if kb:
    lanes["tn_kplanes_bwd"] = kb["WRITE_SIZE_KB"] * 1024 / 4.0
# bytes per training step: every launch of the run (torch glue and the occupancy refresh of the first step included) ...
per_step = sum(v["bytes"] * v["launches_seen"] for v in per.values()) / steps
# ... and the steady state: kernels that ran in every step (the refresh every 64 steps and one-off set-up kernels left out)
per_step_steady = sum(v["bytes"] * v["launches_seen"] for v in per.values() if v["launches_seen"] >= steps) / steps
json.dump({"note": __doc__.strip().replace("\n", " "),
           "command": "scripts/pmc.sh <tag> --steps 3 --warmup 1 --no-stages (one rocprofv3 --pmc pass per counter), then this script",
           "steps_seen": steps, "samples_per_step": samples, "bytes_per_step": per_step_steady, "bytes_per_step_all_launches": per_step, "per_entry": entry,
           "lane_atomics_per_entry": lanes, "per_kernel": per}, open(dst, "w"), indent=1)
print(json.dumps({"per_entry": entry, "lane_atomics_per_entry": lanes, "bytes_per_step": per_step_steady, "bytes_per_step_all_launches": per_step}, indent=1))
