"""Where a wave spends a 32-sample tile of the width-64 heads' forward (mlp_fwd_kernel<..., F2>): s_memtime ticks per phase, summed over
all waves by a library built with -DTN_PHASE_TIMERS (mlp.hip; not the shipped build) and read back through tn_debug_phase_cycles.

    hipcc ... -DTN_PHASE_TIMERS -c tinynerf_amd/csrc/mlp.hip -o mlp_pt.o && hipcc -shared ... -o lib_pt.so <other objects> mlp_pt.o
    TN_LIB_PATH=$PWD/lib_pt.so python scripts/phase_time.py [kp]        (kp: one K-Planes training step as well)

DESIGN 4.2 (round 4) quotes its output: gather 45 %, table rows + first-layer epilogue + stores 26 %, hidden layers 10 % of a tile."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinynerf_amd import models as m, _lib
lib = ctypes.CDLL(os.environ["TN_LIB_PATH"])
def read(reset=1):
    buf = (ctypes.c_ulonglong * 16)()
    lib.tn_debug_phase_cycles(buf, reset)
    return list(buf)
dev = "cuda"
names = {0: "x load + L0 x blocks", 1: "scale", 3: "aux + L0 epilogue + stores", 4: "hidden layer (+ previous stores)", 5: "last stores", 6: "output layer + y", 8: "gather (KP)", 9: "heads (KP tile)"}
def report(tag, n_tiles_waves):
    c = read()
    tot = sum(c)
    print(tag, "total memtime ticks per wave-tile:", tot / max(1, n_tiles_waves))
    for k, v in enumerate(c):
        if v: print(f"   phase {k} {names.get(k, '')}: {v / max(1, n_tiles_waves):.0f} ticks per wave-tile ({100 * v / tot:.1f} %)")
n = 512 * 1024
for name, mod, ind in (("sigma256", m.VanillaOpacityDecoder(256), 256), ("colour256", m.VanillaColorDecoder(8, 256, 64, 3), 256)):
    mod = mod.to(dev)
    x = torch.rand(n, ind, device=dev, requires_grad=True)
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
    args = (x, d) if "colour" in name else (x,)
    y = mod(*args); torch.cuda.synchronize(); read()
    y = mod(*args); torch.cuda.synchronize()
    report(name, n // 32)
if len(sys.argv) > 1:      # K-Planes step
    from tinynerf_amd import rays
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgbs, K, _ = rays.synthetic_scene(n_views=8, res=800, seed=0, device=dev)
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=1024, n_samples=1024, seed=0)
    tr = Trainer(cfg, o, d, rgbs, torch.ones(3, device=dev), torch.device(dev))
    lin = torch.linspace(-1, 1, 128, device=dev)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    tr.occupancy_grid.grid.copy_(torch.where(xx * xx + yy * yy + zz * zz < 0.25, 1.0, tr.occupancy_grid.decay ** 20))
    tr.occupancy_grid.mean = float(tr.occupancy_grid.grid.mean().item())
    tr.occupancy_grid_updates = 10 ** 9
    for _ in range(3): tr.step()
    torch.cuda.synchronize(); read()
    ns = tr.step()["n_samples"]; torch.cuda.synchronize()
    report("kplanes step", int(ns) // 32)
