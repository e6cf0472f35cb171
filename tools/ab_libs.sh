#!/bin/bash
# Alternates bench.py between builds of the library on ONE box:  tools/ab_libs.sh "<bench args>" libA.so libB.so ...
set -e
ARGS=$1; shift
mkdir -p gpurun_out
for round in 1 2; do
  for L in "$@"; do
    DK_LIB_PATH=$PWD/$L timeout -k 10 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-ingest $ARGS > gpurun_out/ab.log 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - "$L" <<'PY'
import json, sys
line = [l for l in open("gpurun_out/ab.log") if l.startswith("{")][0]
d = json.loads(line)
print("%-40s" % sys.argv[1][-40:], round(d["value"], 2), {k: round(v, 2) for k, v in d["stages_ms"].items()},
      "parent", {k: round(v, 2) for k, v in d["parent_build"]["insert_stages_ms"].items()}, flush=True)
PY
  done
done
