"""One forked hipGraph replay of bench.py from a rocprofv3 kernel trace, with the hardware queue of every kernel: the graph executor
spreads a graph over very few queues, and which branch shares a queue with which decides what waits.  Development aid.
usage: graph_queues.py <kernel_trace.csv> [steps back from the last marker, default 19 = a forked replay of `bench.py --steps 10
--warmup 3`; 32 = a single-stream replay of the same run (18 and 31 are the last replays of their forms, followed by the
kernels of the after-timing check (see tools/step_sequence.py for the layout)]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "knn_grid_ranges" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 19
a, b = idx[-2 - back] - 5, idx[-1 - back] - 5        # a replay starts ~5 kernels before its marker (copy, K = 1 search, mesh / stem heads)
t0 = int(rows[a]["Start_Timestamp"])
busy = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    busy[r["Queue_Id"]] = busy.get(r["Queue_Id"], 0) + (e - s)
    print("%9.1f %8.1f q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"], n[:90]))
print("# span %.1f us; busy per queue: %s" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, {k: round(v / 1e3, 1) for k, v in busy.items()}))
