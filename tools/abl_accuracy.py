"""Diagnostic: run the fixture / small oracle parity tests with each variant build (openobj_amd/csrc/abl/lib_*.so)."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sel = sys.argv[1] if len(sys.argv) > 1 else "g5 or eval_points or train_step_vs_oracle or headline"
for so in sorted(glob.glob(os.path.join(root, "openobj_amd/csrc/abl/lib_*.so"))):
    if "PHASE" in so:
        continue
    env = dict(os.environ, OBJNERF_LIB=so)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests/test_hip_parity.py"), "-m", "gpu", "-q",
                          "--tb=line", "-k", sel], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print(os.path.basename(so), "|", out[-1] if out else "no output")
    for l in out:
        if l.startswith("E ") or "AssertionError" in l:
            print("    ", l[:240])
