#!/usr/bin/env python3
"""gpurun_out/<tag>/{FETCH_SIZE,WRITE_SIZE}.json (scripts/pmc.sh) -> profiles/<name>.json: HBM-side bytes per launch and per
C-ABI entry point.  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE tallies a 128-B request of a wide
(16 B/lane) read as 64 B (MI355X_MICROARCH.md, HBM section); every streaming read of these kernels is 16 B/lane."""
import json, sys
src, dst = sys.argv[1], sys.argv[2]
f = json.load(open(f"{src}/FETCH_SIZE.json")); w = json.load(open(f"{src}/WRITE_SIZE.json"))
def short(k): return k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
per = {}
for k in f:
    if not any(t in k for t in ("mlp_", "kplanes_", "sample_", "adam_", "plane_reg", "weights_", "composite_")):
        continue
    per[short(k)] = {"FETCH_SIZE_KB": f[k]["avg"], "WRITE_SIZE_KB": w.get(k, {"avg": 0.0})["avg"], "launches_seen": f[k]["dispatches"],
                     "bytes": (2 * f[k]["avg"] + w.get(k, {"avg": 0.0})["avg"]) * 1024}
def total(*subs):
    return sum(v["bytes"] for k, v in per.items() if any(k.startswith(s) for s in subs))
entry = {
    "tn_mlp_bwd_pair": total("mlp_chain_kernel<64, 4", "mlp_wgrad4_kernel<4", "mlp_wgrad3_kernel<64, 4", "mlp_wgrad_kernel<64, 4", "mlp_wgrad_kernel<64, 1"),
    "tn_mlp_fwd_stash_pair": total("mlp_fwd_kernel<64, true, 16, true, true", "mlp_fwd_kernel<64, true, 12, true, true"),
    "tn_kplanes_bwd": total("kplanes_bwd_kernel"), "tn_kplanes_fwd": total("kplanes_fwd_kernel"),
}
json.dump({"note": __doc__.strip().replace("\n", " "), "command": "scripts/pmc.sh (bench.py --steps 3 --warmup 1 --no-stages, one rocprofv3 --pmc pass per counter)",
           "per_kernel": per, "per_entry": entry}, open(dst, "w"), indent=1)
print(json.dumps(entry, indent=1))
