"""A CU PARTITION for configs[4]'s two kernels, measured: kernel A (issue-bound, fwdr256_kernel) of one chunk on a
stream masked to a subset of the CUs beside kernel B (HBM-bound, wgrad256_kernel) of another chunk on the rest
(hipExtStreamCreateWithCUMask; the library's diagnostic switches OBJ256_ONLY=A|B and OBJ256_NWG).  Prints the time of
each half alone on the whole chip, alone on its partition, and of the pair side by side.

    OBJ256_NWG=<workgroups of kernel A> python tools/c5_cumask.py <CUs for A per 4> [fp16|bf16]
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from openobj_amd import init as obj_init
from openobj_amd import ops, synthetic


def masked_stream(hip, pred):
    words = (C.c_uint32 * 8)()
    for cu in range(256):
        if pred(cu):
            words[cu >> 5] |= 1 << (cu & 31)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    a_of_4 = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mode = sys.argv[2] if len(sys.argv) > 2 else "fp16"
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    hip = C.CDLL("libamdhip64.so")
    K, R, n1, n2, H = 8, 8192, 32, 96, 256
    chunks = []
    for c in range(2):
        arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
        arena.load_stacked(obj_init.init_stacked(K, H, 512, seed=3 + c))
        b = synthetic.random_batch(K, R, n1, n2, seed=11 + c)
        batch = {k: torch.as_tensor(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
        ws = ops.TrainWorkspace(arena, K, R, n1 + n2, False, precision=mode)
        chunks.append((arena, ws, batch))
    if os.environ.get("C5_NO_MASK") == "1":       # no masks: kernel A's OBJ256_NWG persistent workgroups leave the other CUs free
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    else:
        sa = masked_stream(hip, lambda cu: (cu & 3) < a_of_4)
        sb = masked_stream(hip, lambda cu: (cu & 3) >= a_of_4)
    full = torch.cuda.Stream()

    def step(c, only, stream):
        if only:
            os.environ["OBJ256_ONLY"] = only
        else:
            os.environ.pop("OBJ256_ONLY", None)
        arena, ws, batch = chunks[c]
        with torch.cuda.stream(stream):
            ops.train_step(arena, ws, batch, bf16=mode)

    def timeit(fn, reps=5):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    for c in range(2):
        step(c, None, full)
    torch.cuda.synchronize()
    print(f"{mode}, kernel A on {a_of_4}/4 of the CUs (OBJ256_NWG={os.environ.get('OBJ256_NWG', '256')}), kernel B on the rest; 8 objects per half")
    print(f"  A alone, whole chip      {timeit(lambda: step(1, 'A', full)):7.2f} ms")
    print(f"  B alone, whole chip      {timeit(lambda: step(0, 'B', full)):7.2f} ms")
    print(f"  A alone, its partition   {timeit(lambda: step(1, 'A', sa)):7.2f} ms")
    print(f"  B alone, its partition   {timeit(lambda: step(0, 'B', sb)):7.2f} ms")

    def pair():
        step(1, "A", sa)
        step(0, "B", sb)
    print(f"  A and B side by side     {timeit(pair):7.2f} ms")
    os.environ.pop("OBJ256_ONLY", None)


if __name__ == "__main__":
    main()
