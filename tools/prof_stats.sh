#!/bin/bash
# rocprofv3 kernel stats of a python tool, top kernels by total time.  usage: tools/prof_stats.sh tools/bench_pyramid.py [args]   (GPU box)
out=/tmp/prof_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv under", sys.argv[1]); sys.exit(1)
for r in list(csv.DictReader(open(f[0])))[:24]:
    print("%-70s calls %6s avg %9.1f us  total %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"][:5]))
PY
