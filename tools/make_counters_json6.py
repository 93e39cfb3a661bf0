#!/usr/bin/env python3
"""profiles/r06_counters.json (what bench.py reports as RECORDED counter figures) and the profiles/r05_* files from the
PMC / stats output of tools/gpu_profile_round6.sh.     python tools/make_counters_json6.py gpurun_out/prof6 profiles v1"""
import json, os, shutil, sys
src, dst, ver = sys.argv[1:4]
rnd = "r06"

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

def collect(tag, outname):
    """Concatenate the four passes of one workload into profiles/r05_pmc_<outname>_<ver>.txt; returns the merged dict."""
    merged, lines = {}, []
    for p in ("fetch", "write", "sq", "sq2"):
        d = read(f"pmc_{tag}_{p}.txt")
        merged.update(d)
        try:
            lines += [f"# pass {p} (rocprofv3 --kernel-trace --pmc, its own run)\n"] + open(os.path.join(src, f"pmc_{tag}_{p}.txt")).readlines()
        except OSError:
            pass
    if lines:
        open(os.path.join(dst, f"{rnd}_pmc_{outname}_{ver}.txt"), "w").writelines(lines)
    return merged

def hbm(d):
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        return None
    return 2.0 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024

kernels = []
def entry(tag, outname, kernel, K, R=4096, S=64, **extra):
    d = collect(tag, outname)
    if not d:
        return
    e = {"kernel": kernel, "objects": K, "rays": R, "samples": S, "hbm_bytes_per_launch": hbm(d),
         "valu_insts_per_launch": d.get("SQ_INSTS_VALU"), "mfma_insts_per_launch": d.get("SQ_INSTS_MFMA"),
         "lds_insts_per_launch": d.get("SQ_INSTS_LDS"), "lds_bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT"),
         "lds_active_cycles": d.get("SQ_LDS_IDX_ACTIVE"), "wave_cycles_quad": d.get("SQ_WAVE_CYCLES"),
         "wait_any_quad": d.get("SQ_WAIT_ANY"), "mfma_busy_cycles": d.get("SQ_VALU_MFMA_BUSY_CYCLES"),
         "source": f"profiles/{rnd}_pmc_{outname}_{ver}.txt (FETCH_SIZE x 2 + WRITE_SIZE, SQ counters; separate rocprofv3 --pmc passes of bench.py, tools/gpu_profile_round6.sh)"}
    e.update(extra)
    kernels.append(e)

entry("f32", "f32", "train_fused32_kernel<false, false, 64>", 50)
entry("bf16", "bf16", "train_fused_bf16v2_kernel", 50)
entry("c3_f32", "c3_f32", "train_fused32_kernel<true, false, 64>", 50)
entry("c3_bf16", "c3_bf16", "train_fused_bf16v2f_kernel", 50)
entry("c4_f32", "c4share_f32", "train_fused32_kernel<true, false, 64>", 15)
entry("c4_bf16", "c4share_bf16", "train_fused_bf16v2f_kernel", 15)
entry("c4full_f32", "c4_f32", "train_fused32_kernel<true, false, 64>", 120)
entry("c4full_bf16", "c4_bf16", "train_fused_bf16v2f_kernel", 120)
# the background iteration's kernels alone (tools/bg_trace.py --metric: one hidden-128 network, 1200 rays x 64 samples)
entry("bgsmall_f32", "bgsmall_f32", "train_small_kernel<4, false, false>", 1, R=1200)
entry("bgsmall_bf16", "bgsmall_bf16", "train_small_kernel<4, true, false>", 1, R=1200)
entry("bggroup_f32", "bggroup_f32", "gemm_group_kernel", 1, R=1200)
# configs[4]: two kernels per 8-object launch; recorded per object
fw, wg = {}, {}
lines = []
for p in ("fetch", "write", "sq", "sq2"):
    for nm, dd in (("fwd256", fw), ("wgrad256", wg)):
        d = read(f"pmc_c5_{p}_{nm}.txt")
        dd.update(d)
        try:
            lines += [f"# pass {p}, {nm}_kernel\n"] + open(os.path.join(src, f"pmc_c5_{p}_{nm}.txt")).readlines()
        except OSError:
            pass
if fw and wg:
    open(os.path.join(dst, f"{rnd}_pmc_c5_{ver}.txt"), "w").writelines(lines)
    tot = (hbm(fw) or 0) + (hbm(wg) or 0)
    kernels.append({"kernel": "objnerf_train_step, fused hidden-256 path (fwd256_kernel + wgrad256_kernel)", "per_object": True,
                    "objects": 1, "rays": 8192, "samples": 128, "hbm_bytes_per_launch": tot / 8.0,
                    "valu_insts_per_launch": (fw.get("SQ_INSTS_VALU", 0) + wg.get("SQ_INSTS_VALU", 0)) / 8.0,
                    "mfma_insts_per_launch": (fw.get("SQ_INSTS_MFMA", 0) + wg.get("SQ_INSTS_MFMA", 0)) / 8.0,
                    "fwd256_hbm_bytes_per_object": (hbm(fw) or 0) / 8.0, "wgrad256_hbm_bytes_per_object": (hbm(wg) or 0) / 8.0,
                    "wgrad256_lds_bank_conflict_cycles": wg.get("SQ_LDS_BANK_CONFLICT"),
                    "wgrad256_lds_active_cycles": wg.get("SQ_LDS_IDX_ACTIVE"),
                    "source": f"profiles/{rnd}_pmc_c5_{ver}.txt (per object: 8-object launches of bench.py --config c5 --dtype fp16 --objects 8; FETCH_SIZE x 2 + WRITE_SIZE of both kernels, separate --pmc passes)"})
json.dump({"kernels": kernels}, open(os.path.join(dst, f"{rnd}_counters.json"), "w"), indent=1)
for name in ("default", "feat", "c4share", "c5"):
    for fn, out in ((f"stats_{name}/s_kernel_stats.csv", f"{rnd}_kernel_stats_{name}_{ver}.csv"),
                    (f"bench_{name}_under_rocprof.json", f"{rnd}_bench_{name}_under_rocprof_{ver}.json")):
        try:
            shutil.copy(os.path.join(src, fn), os.path.join(dst, out))
        except OSError:
            pass
for tag in ("bg", "bgbf16", "bgfeat", "bgfeatbf16", "bgnative", "bgnativebf16", "bgnativefeat", "bgnativefeatbf16", "mapping"):
    try:
        shutil.copy(os.path.join(src, f"stats_{tag}/s_kernel_stats.csv"), os.path.join(dst, f"{rnd}_kernel_stats_{tag}_{ver}.csv"))
    except OSError:
        pass
for fn, out in (("share_curve.json", f"{rnd}_share_curve_{ver}.json"), ("mapping_bench.txt", f"{rnd}_mapping_bench_{ver}.txt")):
    try:
        shutil.copy(os.path.join(src, fn), os.path.join(dst, out))
    except OSError:
        pass
for name in ("nobg", "c3", "c4", "c5_fp16", "c5_bf16", "dist_selftest", "default_plain", "default_nopipe", "c4share"):
    try:
        shutil.copy(os.path.join(src, f"bench_{name}.json"), os.path.join(dst, f"{rnd}_bench_{name}_{ver}.json"))
    except OSError:
        pass
print(json.dumps({"kernels": [{k: v for k, v in e.items() if k != "source"} for e in kernels]}, indent=1))
