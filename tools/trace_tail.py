"""Last N kernels of a rocprofv3 kernel trace with durations and gaps.  Development aid: trace_tail.py <kernel_trace.csv> [N=40]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 40):]
t0 = int(rows[0]["Start_Timestamp"]); pe = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f gap %7.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0 if pe is None else (s - pe) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:50]))
    pe = e
