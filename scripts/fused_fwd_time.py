#!/usr/bin/env python3
"""A / B: inference forward of the wide stacks, cross-layer persistent launch (csrc/mlp_fused_f2.hip) against one launch per layer
(TN_MLP_LAYERWISE), HIP events on the launch stream, interleaved repetitions.  usage: python scripts/fused_fwd_time.py [n] [reps]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from tinynerf_amd import models          # noqa: E402
from tinynerf_amd.models import _FusedMLP   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = "cuda"
torch.manual_seed(0)
out = {}
import os
only = os.environ.get("FUSED_ONLY", "")
for name, mod, x in (("vanilla_256x10", models.VanillaFeatureMLP(10, 256, 8).to(dev), torch.rand(n, 3, device=dev) * 2 - 1),
                     ("cobafa_128x6", models.MLP(36, 128, 5).to(dev), torch.randn(n, 36, device=dev) * 0.3)):
    if only and only not in name:
        continue
    times = {"fused": [], "layerwise": []}
    with torch.no_grad():
        for r in range(reps + 2):
            for mode in ("fused", "layerwise"):
                _FusedMLP.layerwise_inference = mode == "layerwise"
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                y = mod(x)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times[mode].append(e0.elapsed_time(e1))
                del y
    _FusedMLP.layerwise_inference = False
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    L = 10 if "vanilla" in name else 6
    H = 256 if "vanilla" in name else 128
    flop = 2.0 * n * ((60 if H == 256 else 36) * H + (L - 1) * H * H)
    out[name] = {"n": n, "ms": med, "min_ms": {k: min(v) for k, v in times.items()}, "speedup": med["layerwise"] / med["fused"],
                 "tflops_fp32_equivalent_fused": flop / med["fused"] / 1e9, "frac_of_f16x2_peak_fused": flop / med["fused"] / 1e9 / (2516.0 / 3)}
print(json.dumps(out, indent=1) if not only else json.dumps({k: {"ms": v["ms"], "min": v["min_ms"]} for k, v in out.items()}))
