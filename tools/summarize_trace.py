#!/usr/bin/env python3
"""Steady-state per-kernel summary of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

bench.py's warmup runs MIOpen's find mode (hundreds of trial kernels), so whole-process --stats are
dominated by it.  This tool cuts the trace to the TIMED steps: a step starts at its knn_kernel<1>
launch; the window is the `--steps` timed steps (the instrumented stage-timing steps and the forward in front of the roofline
loop, which follow them, are left out).  Usage:
    python tools/summarize_trace.py gpurun_out/prof/*_kernel_trace.csv --steps 10 > profiles/rNN_steady.csv
"""
import argparse
import collections
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--periodic", action="store_true",
                    help="bench_train.py traces (no kNN launch per step): steps are delimited by a kernel that the trace "
                         "holds exactly once per step")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if a.periodic:
        # a kernel launched exactly once per step (3 warmup + 5 timed steps in bench_train.py) marks the period
        cnt = collections.Counter(r["Kernel_Name"] for r in rows)
        total = a.steps + 3
        marker = next(r["Kernel_Name"] for r in rows if cnt[r["Kernel_Name"]] == total)
        starts = [i for i, r in enumerate(rows) if r["Kernel_Name"] == marker]
        s, e = starts[-(a.steps + 1)], starts[-1]
    else:
        starts = [i for i, r in enumerate(rows) if "knn_kernel<1>" in r["Kernel_Name"]]
        # steps: warmup W, timed K, min(K, 5) instrumented steps (stage events), then one more forward before the roofline loop
        extra = min(a.steps, 5)
        s, e = starts[-(a.steps + 1 + extra)], starts[-(1 + extra)]
    win = rows[s:e]
    t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
    agg = collections.OrderedDict()
    for r in win:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = r["Kernel_Name"]
        x = agg.setdefault(k, [0, 0, 1 << 62, 0])
        x[0] += d
        x[1] += 1
        x[2] = min(x[2], d)
        x[3] = max(x[3], d)
    busy = sum(v[0] for v in agg.values())
    w = csv.writer(sys.stdout)
    w.writerow(["# steady-state window: %d steps, wall %.3f ms/step, GPU busy %.3f ms/step, %d kernels/step" %
                (a.steps, (t1 - t0) / 1e6 / a.steps, busy / 1e6 / a.steps, len(win) // a.steps)])
    w.writerow(["Name", "CallsPerStep", "MsPerStep", "AverageUs", "MinUs", "MaxUs", "PercentOfBusy"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        w.writerow([k, "%.1f" % (v[1] / a.steps), "%.4f" % (v[0] / 1e6 / a.steps), "%.2f" % (v[0] / v[1] / 1e3),
                    "%.2f" % (v[2] / 1e3), "%.2f" % (v[3] / 1e3), "%.2f" % (100.0 * v[0] / busy)])
    # roofline loop kernels (after the window)
    tail = rows[starts[-1]:] if not a.periodic else rows[e:]
    for name in ("match_pipe_sim_kernel", "match_panel_kernel<0, true>", "match_panel_kernel<1, true>", "match_kernel<0, true>", "match_kernel<1, true>"):
        ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail if name in r["Kernel_Name"]]
        if ds:
            w.writerow(["# roofline loop: %s calls=%d avg_us=%.2f min_us=%.2f" % (name, len(ds), sum(ds) / len(ds) / 1e3, min(ds) / 1e3)])


if __name__ == "__main__":
    main()
