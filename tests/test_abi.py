"""CPU checks of the drop-in boundary: the library builds, loads, and exports every symbol that
include/tinynerf_hip.h declares; descriptor structs have the C layout; the product has no CPU path."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tinynerf_hip.h")


@pytest.fixture(scope="module")
def lib():
    from tinynerf_amd import build
    path = build.build(verbose=False)
    return ctypes.CDLL(path)


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tn_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_reference_boundary():
    names = declared_functions()
    assert "tn_weights_fwd" in names and "tn_weights_bwd" in names      # cuda.cu:134-137
    assert len(names) >= 15


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    from tinynerf_amd import _lib as L
    assert lib.tn_abi_version() == L.ABI_VERSION == 6          # (header TN_ABI_VERSION; _lib.lib() refuses any other library)


def test_integration_guide_covers_every_entry_point():
    """INTEGRATION.md names the reference call site behind every entry point the header declares."""
    text = open(os.path.join(os.path.dirname(HEADER), "..", "INTEGRATION.md")).read()
    missing = [n for n in declared_functions() if not re.search(r"\b" + n + r"\b", text)]
    assert not missing, missing


def test_struct_layouts_match_c(tmp_path):
    """sizeof/offsetof of the ctypes mirrors == what a C compiler sees."""
    from tinynerf_amd import _lib as L
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tinynerf_hip.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(tn_sampler_desc), offsetof(tn_sampler_desc, t_table),'
                    'offsetof(tn_sampler_desc, seed), sizeof(tn_mlp_desc), offsetof(tn_mlp_desc, weights),'
                    'offsetof(tn_mlp_desc, biases), sizeof(tn_kplanes_desc), offsetof(tn_kplanes_desc, planes),'
                    'sizeof(tn_adam_reg_item), offsetof(tn_adam_reg_item, cy), offsetof(tn_adam_reg_item, row0), sizeof(tn_adam_item));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(L.SamplerDesc), L.SamplerDesc.t_table.offset, L.SamplerDesc.seed.offset,
            ctypes.sizeof(L.MlpDesc), L.MlpDesc.weights.offset, L.MlpDesc.biases.offset,
            ctypes.sizeof(L.KPlanesDesc), L.KPlanesDesc.planes.offset,
            ctypes.sizeof(L.AdamRegItem), L.AdamRegItem.cy.offset, L.AdamRegItem.row0.offset, ctypes.sizeof(L.AdamItem)]
    assert got == want


def test_argument_errors_are_reported(lib):
    lib.tn_last_error_string.restype = ctypes.c_char_p
    rc = lib.tn_weights_fwd(None, None, None, ctypes.c_float(1e-4), None, ctypes.c_int64(4), ctypes.c_int64(2), None)
    assert rc == -1 and b"null" in lib.tn_last_error_string()
    rc = lib.tn_weights_fwd(None, None, None, ctypes.c_float(1e-4), None, ctypes.c_int64(-1), ctypes.c_int64(2), None)
    assert rc == -2


def test_fused_entry_points_validate_their_configuration(lib):
    """tn_kplanes_mlp_fwd_pair / _bwd_pair / tn_adam_multi_gated / tn_basis_dot_* reject bad arguments before any launch."""
    from tinynerf_amd import _lib as L
    lib.tn_last_error_string.restype = ctypes.c_char_p
    kd, md = L.KPlanesDesc(), L.MlpDesc()
    kd.n_scales, kd.channels = 2, 32                       # the fused launches are built for 3 scales x 32 channels
    md.in_dim = 96
    args = (ctypes.byref(kd), None, ctypes.c_int64(7), ctypes.byref(md), ctypes.byref(md), None, ctypes.c_int64(64), None, None, None,
            None, ctypes.c_int64(0), None, ctypes.c_int64(0), None)
    assert lib.tn_kplanes_mlp_fwd_pair(*args) == -3 and b"3 scales" in lib.tn_last_error_string()
    assert lib.tn_kplanes_mlp_fwd_pair(None, *args[1:]) == -1
    rc = lib.tn_kplanes_mlp_bwd_pair(ctypes.byref(kd), None, ctypes.c_int64(7), None, ctypes.byref(md), ctypes.byref(md), None, None, None, None,
                                     ctypes.c_int64(64), None, None, None, None, None, None, ctypes.c_int64(0), None, ctypes.c_int64(0), None)
    assert rc == -1                                        # null grad_planes
    assert lib.tn_adam_multi_gated(None, ctypes.c_int32(1), *[ctypes.c_float(0.1)] * 5, None, None, ctypes.c_int32(0), None) == -1
    assert lib.tn_basis_dot_fwd(None, None, ctypes.c_int64(8), ctypes.c_int32(96), ctypes.c_int32(5), ctypes.c_int32(0), None, None) == -2
    assert lib.tn_basis_dot_bwd(None, None, None, ctypes.c_int64(0), ctypes.c_int32(96), ctypes.c_int32(3), ctypes.c_int32(2), None, None,
                                ctypes.c_int32(0), None) == 0          # n == 0: nothing to do


def test_no_cpu_path():
    """CPU tensors are rejected like the reference's CHECK_CUDA (cuda.cu:62) -- no fallback."""
    from tinynerf_amd import core
    with pytest.raises(RuntimeError):
        core.NerfWeights.apply(torch.rand(4), torch.rand(4), torch.tensor([[0, 4]], dtype=torch.int32), 1e-4)
    g = core.OccupancyGrid(8, 0.1)
    with pytest.raises(RuntimeError):
        g(torch.zeros(3, 3))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "tinynerf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dirpath, f)


def test_workspace_buckets():
    """host logic: workspace sizes are requested in 1/8-octave steps (models._bucket) -- monotone, never below n, at most 12.5 %
    above it, and a stream of slowly drifting batch sizes maps to a handful of distinct sizes (the caching allocator re-uses
    them instead of growing at every record high)."""
    from tinynerf_amd.models import _bucket
    prev = 0
    for n in list(range(1, 3000)) + [10 ** 6 + 7919 * i for i in range(200)] + [2 ** 20, 2 ** 20 + 1, 2 ** 31 - 1]:
        b = _bucket(n)
        assert b >= n and (b <= n * 1.125 + 64)
        if n > prev:
            assert b >= _bucket(prev) if prev else True
        prev = n
    import random
    rng = random.Random(0)
    sizes = {_bucket(int(1.03e6 * (1 + 0.02 * rng.gauss(0, 1)))) for _ in range(5000)}
    assert len(sizes) <= 3


def test_bench_launcher_parent_is_gpu_free():
    """the launcher branch of bench.py runs before `import torch` and counts GPUs from the KFD topology files"""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    tree = ast.parse(src)
    fn = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    for name in ("launch_ranks", "visible_gpus"):
        used = {n.id for n in ast.walk(fn[name]) if isinstance(n, ast.Name)}
        assert "torch" not in used, name
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and any(a.name.split(".")[0] == "torch" for a in n.names) for n in tree.body)
