"""Where a wave spends a 32-sample tile of the K-Planes chain + scatter launch (mlp_chain_kernel<..., KP>): s_memtime ticks per phase from a
library whose mlp_bwd2.hip was compiled with -DTN_PHASE_TIMERS (see scripts/phase_time.py), read through tn_debug_phase_cycles_b.

    TN_LIB_PATH=$PWD/lib_ptb.so python scripts/phase_time_chain.py

DESIGN 4.2 (round 4) quotes its output: the plane scatter is 82 % of a tile (re-gathering the planes 37 %, run walk + atomics 39 %),
the four-layer chain 9 %."""
import sys, os, ctypes, torch
sys.path.insert(0, os.getcwd())
lib = ctypes.CDLL(os.environ["TN_LIB_PATH"])
def read(reset=1):
    buf = (ctypes.c_ulonglong * 16)()
    lib.tn_debug_phase_cycles_b(buf, reset)
    return list(buf)
from tinynerf_amd import rays
from tinynerf_amd.run import TrainConfig, Trainer
dev = "cuda"
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
ns = int(tr.step()["n_samples"]); torch.cuda.synchronize()
c = read(); tiles = ns // 32
names = {0: "prologue (masks, g_pre, head b G0 + stores)", 1: "chain layers + G row stores", 2: "first-layer dgrad (3 k tiles, both heads)", 3: "scatter scale 0", 4: "scatter scale 1", 5: "scatter scale 2", 8: "  scatter: gather (3 planes) x 3 scales", 9: "  scatter: phase A x 9 planes", 10: "  scatter: phase B (runs + atomics) x 9 planes"}
tot = sum(c[:6])
print("ticks per wave-tile:", tot / tiles)
for k, v in enumerate(c):
    if v: print(f"  {k:2d} {names.get(k,''):55s} {v / tiles:10.0f}  {100 * v / tot:5.1f} %")
