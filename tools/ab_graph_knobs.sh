#!/bin/bash
# A/B of the HIP runtime's graph-executor knobs on the forked hipGraph form of bench.py (development aid).
# usage (GPU box): bash tools/ab_graph_knobs.sh OUTDIR "VAR=VAL" "VAR=VAL VAR2=VAL" ...
out=$1; shift; mkdir -p $out
i=0
for kv in "" "$@"; do
  i=$((i+1))
  env $kv python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 > $out/run_$i.json 2> $out/run_$i.err
  rc=$?
  python - $out/run_$i.json "$kv" $rc <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("[%s] rc=%s" % (sys.argv[2], sys.argv[3]), d["launch_forms"], "forked bit-identical:", d.get("graph_check_forked", {}).get("bit_identical"),
          d.get("graph_check_forked", {}).get("after_timed_replays_bit_identical"), flush=True)
except Exception as e:
    print("[%s] rc=%s no line (%s)" % (sys.argv[2], sys.argv[3], e), flush=True)
P
done
