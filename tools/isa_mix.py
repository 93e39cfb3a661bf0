#!/usr/bin/env python3
"""Instruction mix of one kernel's hottest loop from a hipcc -S listing (CPU-side diagnostic).

usage: isa_mix.py file.s kernel_substring [--all]
Finds the kernel, the backward branch with the largest span (the tile loop) and counts instruction classes in it.
"""
import re, sys, collections

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = i - labels[m.group(1)]
            if best is None or span > best[0]: best = (span, labels[m.group(1)], i)
    lo, hi = (0, len(body)) if "--all" in sys.argv or best is None else (best[1], best[2])
    cnt = collections.Counter(); ops = collections.Counter()
    for l in body[lo:hi]:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m: continue
        op = m.group(1)
        ops[op] += 1
        if op.startswith("v_mfma"): c = "mfma"
        elif op.startswith(("v_sin", "v_cos", "v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): c = "trans"
        elif op.startswith("v_"): c = "valu"
        elif op.startswith("ds_"): c = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
        elif op.startswith("s_waitcnt"): c = "waitcnt"
        elif op.startswith("s_nop"): c = "nop"
        elif op.startswith("s_barrier"): c = "barrier"
        elif op.startswith("s_"): c = "salu"
        else: c = "other"
        cnt[c] += 1
    print(f"loop lines {lo}..{hi} of kernel body ({hi - lo} lines)")
    for k, v in cnt.most_common(): print(f"  {k:8s} {v}")
    print("top ops:")
    for k, v in ops.most_common(45): print(f"  {k:28s} {v}")

main()
