"""Build libtinynerf_hip.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

    python -m tinynerf_amd.build [--force]

hipcc cross-compiles without a GPU.  The built library is git-ignored but travels with the
working tree to the GPU box.  Per-file flags matter: ``sampler.hip`` must keep one IEEE
rounding per operation (``-ffp-contract=off``) to stay bit-exact with the reference's torch
ops, while the MFMA/MLP files want FMA contraction.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libtinynerf_hip.so")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-Wall", "-Wno-unused-function"]
STRICT = ["-ffp-contract=off"]          # one rounding per fp op
FAST = ["-ffp-contract=fast"]
SOURCES = {
    "common.hip": FAST,
    "merge.hip": FAST,
    "heads_dx.hip": FAST,
    "weights.hip": FAST,
    "sampler.hip": STRICT,
    "planes_reg.hip": FAST,
    "kplanes.hip": FAST + ["-munsafe-fp-atomics"],
    "cobafa.hip": STRICT + ["-munsafe-fp-atomics"],   # sawtooth warp x (res-1): an fma in f*x - floor moves taps
    "mlp.hip": FAST,
    "mlp_bwd2.hip": FAST + ["-munsafe-fp-atomics"],
    "mlp_bwd_layers.hip": FAST + ["-munsafe-fp-atomics"],
    "mlp_wgrad_rows.hip": FAST + ["-munsafe-fp-atomics"],
    "mlp_wgrad_rc.hip": FAST + ["-munsafe-fp-atomics"],
    "linear.hip": FAST + ["-munsafe-fp-atomics"],
    "mlp_f2_layers.hip": FAST + ["-munsafe-fp-atomics"],
    "mlp_fused_f2.hip": FAST,
    "mlp_b3_layers.hip": FAST + ["-munsafe-fp-atomics"],
}


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libtinynerf_hip.so cannot be built")


def _digest(paths, flags=None) -> str:
    """file contents AND the effective per-file flag lists: a flag that arrives through the environment (TN_B3_EXTRA_FLAGS) must
    trigger a rebuild like an edit does"""
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode()); h.update(f.read())      # (names, not paths: the tree is copied to the GPU box)
    for k in sorted(flags or {}):
        h.update(("%s: %s\n" % (k, " ".join(COMMON + flags[k]))).encode())
    return h.hexdigest()


def sources():
    return {k: v for k, v in SOURCES.items() if os.path.exists(os.path.join(CSRC, k))}


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sources()
    deps = [os.path.join(CSRC, s) for s in srcs] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "tinynerf_hip.h"))
    deps.append(os.path.abspath(__file__))
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest(deps, srcs)
    # (ablation builds -- TN_B3_ABLATE, TN_ABL_*, TN_FUSED_ABL: wrong results on purpose, for timing -- are made by scripts/build_dev_lib.sh
    #  into a SECOND library; this build has no such switch)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(item):
        src, flags = item
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [hipcc] + COMMON + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs.items()))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
