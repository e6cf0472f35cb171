#!/bin/bash
# On-chip utilisation of a bench workload's kernels (run through gpurun from the repo root):
#   tools/profile_onchip.sh <tag> [bench.py arguments]      e.g.  tools/profile_onchip.sh r02_wgs --workload wgs
# Separate rocprofv3 --pmc passes of derived metrics (never combined with a trace), raw output under
# gpurun_out/prof_<tag>/onchip_*, summary gpurun_out/profiles_<tag>/onchip_pmc.csv (copy it to profiles/<tag>/).
set -e -o pipefail
TAG=${1:-round}
shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT gpurun_out/profiles_$TAG
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-ingest $*"
i=0
for set in "VALUBusy SALUBusy" "VALUUtilization LDSBankConflict" "MemUnitBusy MemUnitStalled" "WriteUnitStalled FetchSize"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/onchip_$i -o onchip -- python3 $ARGS > $OUT/bench_under_onchip_$i.log 2> $OUT/bench_under_onchip_$i.err
    echo "pass $i ($set) done"
done
python3 tools/summarize_onchip.py $OUT gpurun_out/profiles_$TAG/onchip_pmc.csv
