"""Ordered kernel list of ONE step from a rocprofv3 kernel trace: start offset, duration, idle gap before it.  Development aid.
usage: step_sequence.py <kernel_trace.csv> [marker substring, default knn_grid_ranges] [steps back from the last one, default 0]
(bench.py --steps K --warmup W, w = max(3, min(W, 10)) untimed steps in front of each form: from the end 5 instrumented eager steps,
K timed + w eager, K timed + w forked replays (K+w+5 .. 2K+w+4 back), K timed + w single-stream replays (2K+2w+5 .. 3K+2w+4 back))"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "knn_grid_ranges"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
a, b = idx[-2 - back], idx[-1 - back]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
tot = gap_tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    tot += e - s; gap_tot += gap
    print("%9.1f %8.1f %6.1f  st%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r["Stream_Id"], n[:110]))
print("# kernels %d, busy %.1f us, gaps %.1f us, span %.1f us" % (b - a, tot / 1e3, gap_tot / 1e3, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
