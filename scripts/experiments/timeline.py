"""Print the kernel timeline of the LAST msm call from a rocprofv3 kernel-trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last occurrences of k_glv_expand / k_glv_digits (they run side by side) mark the start of the last call
idx = min(max(i for i, r in enumerate(rows) if "k_glv_expand" in r["Kernel_Name"]),
          max(i for i, r in enumerate(rows) if "k_glv_digits" in r["Kernel_Name"]))
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("bp::", "")
    print("%9.1f %9.1f %8.1f  q%-3s %s grid=%s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name, r.get("Grid_Size", "?")))
