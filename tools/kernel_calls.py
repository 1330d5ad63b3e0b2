"""Every dispatch of the kernels matching a substring in a rocprofv3 kernel trace: duration and grid.  Development aid.
usage: kernel_calls.py <kernel_trace.csv> <substring> [last N]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
for r in rows[-n:]:
    print("%8.1f us  grid %s x %s x %s  wg %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"],
                                                 r["Grid_Size_Z"], r["Workgroup_Size_X"]))
