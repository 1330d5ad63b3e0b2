"""Per-kernel average durations from a rocprofv3 kernel trace, in launch order of first appearance, grouped by (name, grid).  Development aid."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, t) in agg.items():
    print("%-92s grid %-8s %-5s n=%-4d avg %.1f us" % (k[0], k[1], k[2], n, t / n))
