"""Print the last step of a rocprofv3 --kernel-trace CSV of tools/bg_trace.py (start offset, duration, queue, grid, name)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "label_counts" in r["Kernel_Name"]]
start = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f +%6.1f us  q%-3s grid %-8s %-5s %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?")[-3:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Kernel_Name"][:70]))
