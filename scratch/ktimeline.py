"""Print the kernel timeline (start offset, duration, gap) of a few steps from a rocprofv3 kernel trace csv dir."""
import csv, glob, sys
d, skip, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:skip + n]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %7.1f  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
    prev_end = max(prev_end, e)
