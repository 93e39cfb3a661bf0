"""Print one step's kernel sequence (start offset, duration, queue, grid, name) from a rocprofv3 --kernel-trace CSV directory.
usage: kernel_timeline.py DIR ANCHOR_KERNEL_SUBSTRING [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
start = idx[-1]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[max(0, start - 3):start + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f +%8.1f us  q%-3s grid %-9s wg %-5s lds %-7s vgpr %-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?")[-3:], r["Grid_Size_X"] + "x" + r["Grid_Size_Y"],
          r["Workgroup_Size_X"], r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), r["Kernel_Name"][:60]))
