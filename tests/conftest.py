"""pytest configuration: the ``gpu`` marker gates everything that needs an MI355X."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    try:            # the CPU port (oracle/torch_port.py) on a 256-thread host: torch's CPU kernels run 140x slower with every hardware
        import torch            # thread than with 32 -- and the checker's time is most of the GPU suite's
        torch.set_num_threads(min(32, os.cpu_count() or 1))
    except Exception:           # noqa: BLE001
        pass


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden


@pytest.fixture(params=["bf16x3", "fp32", "f16x2"])
def matmul(request, monkeypatch):
    """The three matrix modes of the wide stacks (tinynerf_amd.models.MATMUL): "f16x2" -- the default since round 4: two-term fp16
    splits with power-of-two scales on v_mfma_f32_32x32x16_f16 in the forward, data-gradient and weight-gradient layer kernels --,
    "bf16x3" (exact three-way bf16 splits on v_mfma_f32_32x32x16_bf16) and "fp32" (v_mfma_f32_32x32x2_f32).  Every golden / oracle
    comparison that runs a 128- or 256-wide stack takes this fixture, so that ALL kernel families are held against the reference
    directly (round-3 verdict: the fp32-MFMA layer kernels were only covered transitively, HIP against HIP)."""
    from tinynerf_amd import models
    monkeypatch.setattr(models, "MATMUL", request.param)
    return request.param


@pytest.fixture(params=["f16x2", "fp32"])
def heads(request, monkeypatch):
    """Both arithmetic forms of the width-64 heads' forward (mlp_f2_heads.h: f16x2 under MATMUL == "f16x2", the fp32 MFMA otherwise):
    the K-Planes golden / oracle comparisons take this fixture."""
    from tinynerf_amd import models
    monkeypatch.setattr(models, "MATMUL", request.param)
    return request.param
