#!/bin/bash
# rocprofv3 passes for one round; run on the GPU box via gpurun.  Usage: tools/profile.sh <tag> [bench args...]
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2/3: --pmc FETCH_SIZE / WRITE_SIZE in their own runs; passes 4/5: SQ counters
# The PMC passes render 32 spp = one full default batch per launch, so their per-launch bytes match the timed run's launches.
# (TCC slots do not fit both; never combined with other trace domains).
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 32 "$@" > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 32 "$@" > $OUT/bench_pmc_write.log 2>&1
# Passes 4/5 (SQ instruction counts and utilisation; each its own --pmc run)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $OUT/pmc_inst -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 32 "$@" > $OUT/bench_pmc_inst.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_util -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 32 "$@" > $OUT/bench_pmc_util.log 2>&1
find $OUT -name "*.csv" | head -40
tail -2 $OUT/bench_trace.log
