"""Diagnostic: time of one background-network step (hidden 128, 1200 rays x 64 samples: the bench's background batch)
in fp32 and bf16 mode for every variant build in openobj_amd/csrc/abl (OBJNERF_LIB)."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time, torch
sys.path.insert(0, %r)
from openobj_amd import ops, synthetic, init as obj_init
dev = torch.device("cuda:0")
arena = ops.ParamArena(1, ops.NetShape(128, 512, 6), dev)
arena.load_stacked(obj_init.init_stacked(1, 128, 512, seed=1))
b = synthetic.random_batch(1, 1200, 16, 48, seed=1)
batch = {k: torch.from_numpy(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
ws = ops.TrainWorkspace(arena, 1, 1200, 64, False)
for mode in (False, True):
    for _ in range(10): ops.train_step(arena, ws, batch, bf16=mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ops.train_step(arena, ws, batch, bf16=mode)
    torch.cuda.synchronize(); print("bf16" if mode else "fp32", "%%.3f ms" %% ((time.perf_counter() - t0) / 50 * 1e3), end="  ")
print()
''' % root
for so in sorted(glob.glob(os.path.join(root, "openobj_amd/csrc/abl/lib_*.so"))):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OBJNERF_LIB=so), capture_output=True, text=True)
    print(os.path.basename(so), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
