"""One hidden-256 step (no feature loss) through the fused kernels; writes grads + loss terms to an .npz.
Run twice -- as is (row-split kernel A) and with OBJ256_FIRST_FORM=1 (fwd256_kernel) -- to compare the two forms
(tests/test_fp16_gpu.py::test_hidden256_kernel_forms_agree)."""
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from openobj_amd import init as obj_init
from openobj_amd import ops, synthetic


def main():
    out, mode = sys.argv[1], sys.argv[2]
    K, R, n1, n2 = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    dev = torch.device("cuda:0")
    arena = ops.ParamArena(K, ops.NetShape(256, 512, 6), dev)
    arena.load_stacked(obj_init.init_stacked(K, 256, 512, seed=29))
    b = synthetic.random_batch(K, R, n1, n2, seed=31)
    batch = {k: torch.as_tensor(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, False, precision=mode)
    ops.train_step(arena, ws, batch, bf16=mode)
    torch.cuda.synchronize()
    np.savez(out, grads=ws.grads.cpu().numpy(), terms=ws.loss_terms.cpu().numpy(), status=int(ws.status.item()))


if __name__ == "__main__":
    main()
