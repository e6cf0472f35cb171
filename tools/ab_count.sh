#!/bin/bash
# Alternates bench.py (timed steps + end-to-end counting pass) between builds of the library on ONE box:
#   tools/ab_count.sh libA.so libB.so ...
set -e
mkdir -p gpurun_out
for round in 1 2; do
  for L in "$@"; do
    DK_LIB_PATH=$PWD/$L timeout -k 10 400 python bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-other-workloads --no-ingest > gpurun_out/abc.log 2> gpurun_out/abc.err || { tail -5 gpurun_out/abc.err; exit 1; }
    python - "$L" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/abc.log") if l.startswith("{")][0])
e = d.get("end_to_end", {})
print("%-28s" % sys.argv[1][-28:], round(d["value"], 2), {k: round(v, 2) for k, v in d["stages_ms"].items()},
      "e2e counting", e.get("counting_stages_ms"), "child-only", e.get("child_only_kmers"), flush=True)
PY
  done
done
