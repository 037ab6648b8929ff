import csv, glob, collections, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X"), r.get("VGPR_Count"), r.get("LDS_Block_Size"))
            d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        v = sorted(v)
        print(k, "n", len(v), "median_ns", v[len(v) // 2], "min", v[0], "mean", sum(v) // len(v))
