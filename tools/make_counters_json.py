#!/usr/bin/env python3
"""profiles/rNN_counters.json (what bench.py reports as RECORDED counter figures) from the PMC summaries of
tools/gpu_profile_round.sh.    python tools/make_counters_json.py gpurun_out/prof profiles r02 v15"""
import json, os, sys
src, dst, rnd, ver = sys.argv[1:5]

def read(name):
    out = {}
    try:
        for l in open(os.path.join(src, name)):
            p = l.split()
            if len(p) >= 3 and p[-1].startswith("mean="):
                out[p[0]] = float(p[-1][5:])
    except OSError:
        pass
    return out

f, w, sq, sq2, b1 = read("pmc_fetch.txt"), read("pmc_write.txt"), read("pmc_sq.txt"), read("pmc_sq2.txt"), read("pmc_bf16.txt")
kernels = []
if f and w:
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)
    hbm = 2.0 * f["FETCH_SIZE"] * 1024 + w["WRITE_SIZE"] * 1024
    kernels.append({"kernel": "train_fused32_kernel<false, false, 64>", "objects": 50, "rays": 4096, "samples": 64,
                    "hbm_bytes_per_launch": hbm, "valu_insts_per_launch": sq.get("SQ_INSTS_VALU"),
                    "mfma_insts_per_launch": sq2.get("SQ_INSTS_MFMA"), "lds_insts_per_launch": sq2.get("SQ_INSTS_LDS"),
                    "source": f"profiles/{rnd}_pmc_hbm_{ver}.txt, profiles/{rnd}_pmc_sq_{ver}.txt (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes)"})
if b1:
    kernels.append({"kernel": "train_fused_bf16_kernel<false, 64>", "objects": 50, "rays": 4096, "samples": 64,
                    "valu_insts_per_launch": b1.get("SQ_INSTS_VALU"), "mfma_insts_per_launch": b1.get("SQ_INSTS_MFMA"),
                    "lds_insts_per_launch": b1.get("SQ_INSTS_LDS"), "hbm_bytes_per_launch": None,
                    "source": f"profiles/{rnd}_pmc_bf16_{ver}.txt (SQ_INSTS_VALU, --pmc pass of bench.py --dtype bf16 --no-bg)"})
json.dump({"kernels": kernels}, open(os.path.join(dst, f"{rnd}_counters.json"), "w"), indent=1)
print(json.dumps({"kernels": kernels}, indent=1))
