#!/bin/bash
# The rocprofv3 evidence of one round; run on the GPU box via gpurun.  Usage: tools/profile.sh <tag> [bench args...]
#  pass 1  rocprofv3 --kernel-trace --stats around one frame of bench.py               -> per-kernel time
#  pass 2  bench.py itself (the driver's command), which runs its own rocprofv3 --pmc child passes (FETCH_SIZE | WRITE_SIZE | SQ
#          counters, each group its own run, never combined with a trace domain) and keeps their output   -> counters + the bench line
# tools/summarize_profile.py <tag> then turns gpurun_out/prof_<tag>/ into the tracked files under profiles/.
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-drop-in "$@" > $OUT/bench_trace.log 2>&1
cd $REPO
python3 bench.py --steps 3 --warmup 1 --pmc-keep $OUT/pmc --pmc-save $TAG "$@" > $OUT/bench.json 2> $OUT/bench.err
cp profiles/pmc_kernels*.json $OUT/ 2>/dev/null   # (bench.py --pmc-save wrote the file of THIS workload: pmc_kernels.json for the default one)
find $OUT -name "*.csv" | head -20
tail -2 $OUT/bench_trace.log | cut -c1-400
cat $OUT/bench.json
