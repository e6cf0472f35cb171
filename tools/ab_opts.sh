#!/bin/bash
# Alternates bench.py between engine-option sets on ONE box (boxes differ by +-3 %):
#   tools/ab_opts.sh "<bench args>" "<opts A>" "<opts B>" ...      e.g.  tools/ab_opts.sh "--reads 256000000" "" "--opt scan_bits=9"
# prints value + child stage times + parent stage times of every run, two rounds each
set -e
ARGS=$1; shift
mkdir -p gpurun_out
for round in 1 2; do
  for O in "$@"; do
    timeout -k 10 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-ingest $ARGS $O > gpurun_out/ab.log 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - "$O" <<'PY'
import json, sys
line = [l for l in open("gpurun_out/ab.log") if l.startswith("{")][0]
d = json.loads(line)
print("[%s]" % sys.argv[1], round(d["value"], 2), {k: round(v, 2) for k, v in d["stages_ms"].items()},
      "parent", round(d["parent_build"]["insert_ms_per_batch"], 2), {k: round(v, 2) for k, v in d["parent_build"]["insert_stages_ms"].items()}, flush=True)
PY
  done
done
