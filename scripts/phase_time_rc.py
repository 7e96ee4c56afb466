"""Where a wave spends a 32-sample tile of wgrad_rc_kernel (csrc/mlp_wgrad_rc.hip, TN_MLP_LEAN): s_memtime ticks per phase from a library
whose mlp_wgrad_rc.hip was compiled with -DTN_PHASE_TIMERS, read through tn_debug_phase_cycles_rc.

    scripts/build_dev_lib.sh scratch/lib_ptrc.so mlp_wgrad_rc.hip -DTN_PHASE_TIMERS
    TN_LIB_PATH=$PWD/scratch/lib_ptrc.so python scripts/phase_time_rc.py
"""
import sys, os, ctypes, torch
sys.path.insert(0, os.getcwd())
lib = ctypes.CDLL(os.environ["TN_LIB_PATH"])
def read(reset=1):
    buf = (ctypes.c_ulonglong * 16)()
    lib.tn_debug_phase_cycles_rc(buf, reset)
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
names = {0: "layer 0, table columns (S + F) + epilogue", 1: "x rows -> fp16 operands (tile maximum, split)", 2: "layer 0, x columns (S + F) + epilogue",
         3: "sigma head: layer 0 (F), dW_1s on the VALU", 4: "dW_1 (splits of G_1 / H_1, 48 bf16 MFMAs)", 5: "dW_2", 6: "dW_3",
         7: "layer 1 (S + F) + epilogue", 8: "layer 2 (S + F) + epilogue", 9: "layer 3 (F) + epilogue", 10: "output layer: dW_4 on the VALU"}
tot = sum(c[:11])
print("ticks per wave-tile:", tot / tiles, "(100 MHz timer ticks x 21 = ~cycles at 2.1 GHz)" )
for k, v in enumerate(c[:11]):
    print(f"  {k:2d} {names.get(k,''):55s} {v / tiles:10.1f}  {100 * v / tot:5.1f} %")
