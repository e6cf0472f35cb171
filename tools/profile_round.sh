#!/bin/bash
# Profiles a bench workload on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [bench.py arguments]      e.g.  tools/profile_round.sh r02_wgs --workload wgs
# Three separate rocprofv3 runs of the same command (kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE --
# the two TCC counters do not fit one pass and --pmc is never combined with other traces), raw output
# under gpurun_out/prof_<tag>/, summaries written by tools/summarize_profiles.py into gpurun_out/profiles_<tag>/
# (copy those to profiles/<tag>/ and merge traffic.json into profiles/traffic.json to have them tracked).
set -e -o pipefail
TAG=${1:-round}
shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# BENCH: another driver script (tools/kmers_bench.py); default = bench.py in its profiling shape
if [ -n "$BENCH" ]; then ARGS="$BENCH $*"; else ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-ingest $*"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ARGS > $OUT/bench_under_stats.log 2> $OUT/bench_under_stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ARGS > $OUT/bench_under_fetch.log 2> $OUT/bench_under_fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ARGS > $OUT/bench_under_write.log 2> $OUT/bench_under_write.err
echo "WRITE_SIZE pass done"
python3 tools/summarize_profiles.py $OUT gpurun_out/profiles_$TAG
