"""Every switch that selects a code path, read from the environment ONCE (at import) into one object.

The product has one arithmetic by default -- "f16x2": matrix products as two-term fp16 splits with power-of-two scales on the fp16 matrix
cores, fp32 storage / accumulation / results -- and two others kept as parity partners of the test matrix ("bf16x3": exact three-term
bf16 splits; "fp32": v_mfma_f32_32x32x2_f32).  The remaining switches turn single optimisations of the f16x2 path off again for A / B timing
and for the tests that hold an optimised form against its plain one.  They are only honoured together with ``matmul == "f16x2"``: with
the other two arithmetics they must stay at their defaults (``Config.validate``), so that no combination exists that no test runs
(``supported_combinations`` is what tests/test_config.py walks).

    TN_MATMUL          f16x2 | bf16x3 | fp32     arithmetic of the wide stacks and of the heads' forward (models.MATMUL)
    TN_ROWS_HANDOFF    1 | 0    heads read a wide stack's output from its workspace rows, the row-major copy is not written (fused.ROWS_HANDOFF)
    TN_MERGE_LAST      1 | 0    the stack's last Linear merged into the heads' first layers, TN_MLP_SKIP_LAST (fused.MERGE_LAST); needs ROWS_HANDOFF
    TN_HEADS_PAIR      1 | 0    both heads' first-layer weight gradients / d loss / d x behind a wide stack as joint launches (fused.HEADS_PAIR_BACKWARD)
    TN_KP_LEAN         1 | 0    K-Planes heads: no hidden activations stashed, rebuilt in the weight-gradient launches, TN_MLP_LEAN (fused.KP_LEAN)
    TN_INFER_PAIR      1 | 0    inference: gather + both heads of every sample in one launch while most samples are live (fused.INFER_PAIR)
    TN_ADAM_OVERLAP    1 | 0    N == 1: the planes' optimizer pass on a side stream beside the weight-gradient kernels (run.ADAM_OVERLAP)
    TN_SIDE_PLAN       1 | 0    the next step's sampler pass on a stream of its own (run.Trainer)

Not switches of the computation, read by ``_lib``: TN_LIB_PATH (load another build of the library: A / B of two builds on one box) and
TN_TRACE (synchronise after every launch and print its name: fault bisection).

Build-time ablation macros (TN_F2_NT, TN_ABL_*, TN_B3_ABLATE, TN_FUSED_ABL) are not here: only scripts/build_dev_lib.sh sets them, into a
second library that computes wrong results on purpose and is never the one ``_lib`` loads by default.
"""
from __future__ import annotations

import itertools
import os
from dataclasses import dataclass, fields, replace
from typing import Iterator, Mapping

MATMULS = ("f16x2", "bf16x3", "fp32")
_FLAGS = {"rows_handoff": "TN_ROWS_HANDOFF", "merge_last": "TN_MERGE_LAST", "heads_pair": "TN_HEADS_PAIR", "kp_lean": "TN_KP_LEAN",
          "infer_pair": "TN_INFER_PAIR", "adam_overlap": "TN_ADAM_OVERLAP", "side_plan": "TN_SIDE_PLAN"}


@dataclass(frozen=True)
class Config:
    matmul: str = "f16x2"
    rows_handoff: bool = True
    merge_last: bool = True
    heads_pair: bool = True
    kp_lean: bool = True
    infer_pair: bool = True
    adam_overlap: bool = True
    side_plan: bool = True

    def validate(self) -> "Config":
        if self.matmul not in MATMULS:
            raise RuntimeError(f"TN_MATMUL={self.matmul}: fp32, bf16x3 or f16x2")
        off = [n for n in _FLAGS if not getattr(self, n)]
        if self.matmul != "f16x2" and off:
            raise RuntimeError(f"tinynerf_amd.config: {', '.join(_FLAGS[n] + '=0' for n in off)} with TN_MATMUL={self.matmul}: the A / B switches "
                               "belong to the f16x2 path; with bf16x3 / fp32 they stay at their defaults (see tinynerf_amd/config.py)")
        if self.merge_last and not self.rows_handoff:
            # (not an error: without the row handoff there is nothing to merge into -- the merge is simply not armed; normalised so that
            #  two spellings of one behaviour are one configuration)
            return replace(self, merge_last=False)
        return self

    @staticmethod
    def from_env(env: Mapping[str, str] = os.environ) -> "Config":
        kw = {"matmul": env.get("TN_MATMUL", "f16x2").lower()}
        for name, var in _FLAGS.items():
            kw[name] = env.get(var, "1") != "0"
        return Config(**kw).validate()

    def as_env(self) -> dict:
        return {"TN_MATMUL": self.matmul, **{var: "1" if getattr(self, name) else "0" for name, var in _FLAGS.items()}}


def supported_combinations() -> Iterator[Config]:
    """every configuration ``validate`` accepts, each behaviour once: the three arithmetics at their defaults, and for f16x2 every setting
    of the A / B switches (merge_last only with rows_handoff)"""
    seen = set()
    for mm in MATMULS:
        names = list(_FLAGS) if mm == "f16x2" else []
        for bits in itertools.product((True, False), repeat=len(names)):
            try:
                c = Config(matmul=mm, **dict(zip(names, bits))).validate()
            except RuntimeError:
                continue
            if c not in seen:
                seen.add(c)
                yield c


def apply(cfg: Config) -> None:
    """set the module-level switches of an already imported package to `cfg` (tests; the product reads CONFIG once at import)"""
    from . import fused, models, run
    cfg = cfg.validate()
    models.MATMUL = cfg.matmul
    fused.ROWS_HANDOFF, fused.MERGE_LAST, fused.HEADS_PAIR_BACKWARD = cfg.rows_handoff, cfg.merge_last, cfg.heads_pair
    fused.KP_LEAN, fused.INFER_PAIR = cfg.kp_lean, cfg.infer_pair
    run.ADAM_OVERLAP, run.SIDE_PLAN = cfg.adam_overlap, cfg.side_plan


CONFIG = Config.from_env()
