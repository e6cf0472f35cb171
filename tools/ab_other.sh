#!/bin/bash
# Alternates a short bench.py run of another workload (chr20 / ont) between builds of the library on ONE box:
#   tools/ab_other.sh <workload> libA.so libB.so ...
set -e
W=$1; shift
mkdir -p gpurun_out
for round in 1 2; do
  for L in "$@"; do
    DK_LIB_PATH=$PWD/$L timeout -k 10 300 python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/abo.log 2> gpurun_out/abo.err || { tail -5 gpurun_out/abo.err; exit 1; }
    python - "$L" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/abo.log") if l.startswith("{")][0])
print("%-24s" % sys.argv[1][-24:], round(d["value"], 2), {k: round(v, 2) for k, v in d["stages_ms"].items()},
      "absent", d["pass_stats"]["n_absent"], "emitted", d["pass_stats"].get("n_emitted"), flush=True)
PY
  done
done
