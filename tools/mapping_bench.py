"""Where does a frame of the mapping loop (openobj_amd/mapping.py) spend its time at the reference's native
shape?  Synthetic 1200 x 680 frames, N objects on a grid in front of a wall, room_0 hyper-parameters
(100 iterations x 120 rays x 10 samples per object, background 1200 rays x 14 samples).  Run on the GPU box."""
import argparse
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openobj_amd import cfg as ocfg, mapping

ap = argparse.ArgumentParser()
ap.add_argument("--objects", type=int, default=50)
ap.add_argument("--frames", type=int, default=6)
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--part", action="store_true")
ap.add_argument("--only", choices=["obj", "bg"], default=None, help="time only the object or only the background steps")
ap.add_argument("--overlap", action="store_true", help="background steps on a second stream")
a = ap.parse_args()
dev = "cuda:0"
W, H = 1200, 680
c = ocfg.Config(ocfg.replica_room0_config(train_device=dev, **{"trainer.part_mode": int(a.part)}))
from openobj_amd import synthetic


def sample(i):
    return synthetic.grid_frame(i, a.objects, W, H, a.part)


m = mapping.IncrementalMapper(c, bf16=a.bf16)
sync = torch.cuda.synchronize
for i in range(a.frames):
    s = sample(i)
    sync(); t0 = time.perf_counter()
    m.ingest(s, i)
    sync(); t1 = time.perf_counter()
    if a.only is None and not a.overlap:
        # the mapper's own frame: pools, iterations (background steps on a second stream), copy-back
        t_s = time.perf_counter()
        m._ensure_stack()
        pool, bg_pool = m.sample_pools()
        sync(); t2 = time.perf_counter()
        real_sample = m.sample_pools
        m.sample_pools = lambda: (pool, bg_pool)
        m.train_frame()
        m.sample_pools = real_sample
        sync(); t3 = time.perf_counter()
        t4 = t3
        print("frame %d: ingest %.1f ms | stack+sample pools %.1f ms | %d iterations + copy-back %.1f ms (%.3f ms each)"
              % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t1), c.n_iter_per_frame, 1e3 * (t3 - t2), 1e3 * (t3 - t2) / c.n_iter_per_frame),
              flush=True)
        continue
    m._ensure_stack()
    pool, bg_pool = m.sample_pools()
    sync(); t2 = time.perf_counter()
    npo, npo_bg = c.n_per_optim, c.n_per_optim_bg
    pool = {k: v.reshape(v.shape[0], c.n_iter_per_frame, npo, *v.shape[2:]).transpose(0, 1).contiguous()
            for k, v in pool.items()}                   # as IncrementalMapper.train_frame: [n_iter, K, npo, ...]
    if a.overlap:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for it in range(c.n_iter_per_frame):
                m.bg_loop.step({k: v[:, it * npo_bg:(it + 1) * npo_bg].contiguous() for k, v in bg_pool.items()})
        for it in range(c.n_iter_per_frame):
            m.loop.step({k: v[it] for k, v in pool.items()})
        torch.cuda.current_stream().wait_stream(side)
    else:
        for it in range(c.n_iter_per_frame):
            if a.only != "bg":
                m.loop.step({k: v[it] for k, v in pool.items()})
            if a.only != "obj":
                m.bg_loop.step({k: v[:, it * npo_bg:(it + 1) * npo_bg].contiguous() for k, v in bg_pool.items()})
    sync(); t3 = time.perf_counter()
    m.loop.copy_back()
    sync(); t4 = time.perf_counter()
    print("frame %d: ingest %.1f ms | stack+sample pools %.1f ms | %d iterations %.1f ms (%.3f ms each) | copy-back %.1f ms"
          % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t1), c.n_iter_per_frame, 1e3 * (t3 - t2), 1e3 * (t3 - t2) / c.n_iter_per_frame,
             1e3 * (t4 - t3)), flush=True)
print("objects:", len(m.obj_dict), "| HBM in use: %.1f GB" % (torch.cuda.memory_allocated() / 2**30))
