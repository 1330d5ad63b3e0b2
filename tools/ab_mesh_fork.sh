#!/bin/bash
# A/B of where the mesh fork is enqueued in the forked hipGraph form (settings.MESH_FORK_AT; -1 = behind the embedding).
# usage (GPU box): bash tools/ab_mesh_fork.sh OUTDIR
out=${1:-gpurun_out/meshfork}; mkdir -p $out
for k in -1 0 1 2 3; do
  GDM_MESH_FORK_AT=$k python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 > $out/at_$k.json 2> $out/at_$k.err || exit 1
  python - $out/at_$k.json $k <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("MESH_FORK_AT=%s" % sys.argv[2], d["launch_forms"], "forked bit-identical:", d.get("graph_check_forked", {}).get("bit_identical"),
      d.get("graph_check_forked", {}).get("after_timed_replays_bit_identical"))
P
done
