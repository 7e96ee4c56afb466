#!/usr/bin/env python3
"""gpurun_out/<tag>/sq.json or pmc.json (scripts/pmc_sq.sh / pmc_any.sh with the eight SQ counters below) -> one profile file:
per kernel the share of wave cycles parked in s_waitcnt / barrier (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY) and issuing
(SQ_ACTIVE_INST_ANY).  usage: scripts/sq_to_json.py <out.json> <note> <in.json> [<in.json> ...]"""
import json
import sys

out, note, srcs = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for src in srcs:
    d = json.load(open(src))
    for k, v in d.items():
        if not isinstance(v, dict) or not v.get("SQ_WAVE_CYCLES"):
            continue
        w = v["SQ_WAVE_CYCLES"]
        v = dict(v)
        v["fractions_of_wave_cycles"] = {m: round(v.get(m, 0.0) / w, 3) for m in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
        res[k] = v
keep = sorted(res, key=lambda k: -res[k]["SQ_WAVE_CYCLES"] * res[k].get("launches", 1))[:24]
json.dump({"note": note, "per_kernel": {k: res[k] for k in keep}}, open(out, "w"), indent=1)
for k in keep[:12]:
    print(k[:70], res[k]["fractions_of_wave_cycles"])
