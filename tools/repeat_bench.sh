#!/bin/bash
# Runs bench.py N times in fresh processes on one box and prints the stage times of each run (run-to-run spread):
#   tools/repeat_bench.sh N "<bench args>"
N=$1; shift
mkdir -p gpurun_out
for i in $(seq 1 "$N"); do
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-other-workloads --no-ingest $1 > gpurun_out/rep.log 2> gpurun_out/rep.err || { tail -3 gpurun_out/rep.err; exit 1; }
  python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/rep.log") if l.startswith("{")][0])
print(round(d["value"], 2), {k: round(v, 2) for k, v in d["stages_ms"].items()}, flush=True)
PY
done
