"""Diagnostic: time bench.py with each variant build in openobj_amd/csrc/abl/lib_*.so (OBJNERF_LIB override).
Builds come from tools/build_variant.sh / build_variant_generic.sh (e.g. -DPHASE_TIMING, -DOBJ_GEMM_BK=32).
Extra arguments go to bench.py."""
import glob, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = sys.argv[1:] or ["--no-bg"]
for so in sorted(glob.glob(os.path.join(root, "openobj_amd/csrc/abl/lib_*.so"))):
    env = dict(os.environ, OBJNERF_LIB=so)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "2",
                          "--no-cpu-baseline", "--no-bf16-line"] + extra, env=env, capture_output=True,
                         text=True).stdout.strip().splitlines()
    try:
        d = json.loads(out[-1])
        print(os.path.basename(so), "ms_per_step %.3f kernel_ms %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]))
    except Exception as e:
        print(os.path.basename(so), "FAILED", out[-3:] if out else e)
