#!/bin/bash
# Alternates bench.py between two builds of the library on ONE box (boxes differ by +-3 %, more than most
# kernel tweaks are worth):  tools/ab_bench.sh path/to/libA.so path/to/libB.so [bench.py args...]
# Typical use from the CPU container:
#   git stash; make -C denovo_kmer_amd/csrc; cp denovo_kmer_amd/libdenovo_kmer.so tools/experiments/libdk_base.so; git stash pop
#   make -C denovo_kmer_amd/csrc
#   gpurun -- 'tools/ab_bench.sh tools/experiments/libdk_base.so denovo_kmer_amd/libdenovo_kmer.so'
set -e
A=$1; B=$2; shift 2
mkdir -p gpurun_out
for round in 1 2; do
  for L in "$A" "$B"; do
    DK_LIB_PATH=$PWD/$L timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/ab.log 2>&1
    python - "$L" <<'PY'
import json, sys
line = [l for l in open("gpurun_out/ab.log") if l.startswith("{")][0]
d = json.loads(line)
print(sys.argv[1][-28:], round(d["value"], 2), {k: round(v, 2) for k, v in d["stages_ms"].items()},
      {k: round(v, 2) for k, v in d["parent_build"]["insert_stages_ms"].items()})
PY
  done
done
