#!/usr/bin/env python3
"""Instruction classes per phase of a -DPHASE_TIMING build listing (hipcc -S): segments between consecutive
s_memtime stamps inside the tile loop of the named kernel.  usage: isa_phases.py file.s kernel_substring"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
seg, segs = collections.Counter(), []
for l in body:
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if not m: continue
    op = m.group(1)
    if op == "s_memtime":
        segs.append(seg); seg = collections.Counter(); continue
    if op.startswith("v_mfma"): c = "mfma"
    elif op.startswith(("v_sin", "v_cos", "v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): c = "trans"
    elif "dpp" in l and op.startswith("v_"): c = "dpp"
    elif op.startswith(("v_cndmask", "v_cmp")): c = "sel"
    elif op.startswith("v_"): c = "valu"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
    elif op.startswith("s_waitcnt"): c = "wait"
    elif op.startswith("s_nop"): c = "nop"
    elif op.startswith("s_barrier"): c = "bar"
    elif op.startswith("s_"): c = "salu"
    else: c = "other"
    seg[c] += 1
segs.append(seg)
cols = ["mfma", "valu", "sel", "dpp", "trans", "lds", "vmem", "salu", "wait", "nop", "bar"]
print("seg " + " ".join("%6s" % c for c in cols))
for i, s in enumerate(segs):
    print("%3d " % i + " ".join("%6d" % s[c] for c in cols))
