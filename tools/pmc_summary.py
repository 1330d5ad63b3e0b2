"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter).  Development aid.
usage: pmc_summary.py <counter_collection.csv> <kernel-name substring> [more substrings]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for pat in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for r in rows:
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("## %s" % pat)
    for k in sorted(acc):
        v = acc[k]
        print("%-32s %16.0f   (%d dispatches)" % (k, sum(v) / len(v), len(v)))
