"""Diagnostic: time the fused kernel of ablation builds (openobj_amd/csrc/abl/lib_*.so).  Outputs are WRONG in
the ablated builds; only the time matters."""
import os, sys, glob, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for so in sorted(glob.glob(os.path.join(root, "openobj_amd/csrc/abl/lib_*.so"))):
    env = dict(os.environ, OBJNERF_LIB=so)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-bg", "--steps", "8", "--warmup", "2",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    import json
    try:
        d = json.loads(out[-1])
        print(os.path.basename(so), "kernel_ms %.2f" % d["roofline"]["kernel_ms"])
    except Exception as e:
        print(os.path.basename(so), "FAILED", out[-3:] if out else e)
