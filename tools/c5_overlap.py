"""The rejected overlap of configs[4]'s two kernels, measured: chunk i's weight-gradient kernel (wgrad256_kernel,
HBM-bound) beside chunk i + 1's kernel A (MFMA / issue-bound), by issuing two chunks' steps on two streams.  Both kernels
hold ~150 KB of LDS per workgroup, so no workgroup of one fits on a CU that runs the other: the hardware can only fill
the CUs one kernel's tail leaves idle.  Prints the time of two 8-object chunks issued back to back on ONE stream and on
TWO streams.

    python tools/c5_overlap.py [fp16|bf16]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from openobj_amd import init as obj_init
from openobj_amd import ops, synthetic


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    dev = torch.device("cuda:0")
    K, R, n1, n2, H = 8, 8192, 32, 96, 256
    chunks = []
    for c in range(2):
        arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
        arena.load_stacked(obj_init.init_stacked(K, H, 512, seed=3 + c))
        b = synthetic.random_batch(K, R, n1, n2, seed=11 + c)
        batch = {k: torch.as_tensor(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
        ws = ops.TrainWorkspace(arena, K, R, n1 + n2, False, precision=mode)
        chunks.append((arena, ws, batch))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(two_streams, reps=6):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            for c, (arena, ws, batch) in enumerate(chunks):
                with torch.cuda.stream(streams[c if two_streams else 0]):
                    ops.train_step(arena, ws, batch, bf16=mode)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    run(False, 2); run(True, 2)
    for _ in range(2):
        a, b = run(False), run(True)
        print(f"{mode}: two 8-object chunks, one stream {a:.2f} ms, two streams {b:.2f} ms ({(1 - b / a) * 100:+.1f} % saved)")


if __name__ == "__main__":
    main()
