#!/bin/bash
# Diagnostic counter passes over the three hot kernels (run on the GPU box): tools/pmc_diag.sh <scene> <outdir> [bench args]
# One rocprofv3 --pmc run per counter group (never combined with a trace domain), each over one full-batch frame of bench.py;
# prints, per kernel, the per-launch average of every counter.  Used for DESIGN.md section 4 "what bounds it".
# (No TA_* / TCP_* groups: TA_BUSY_avr + TA_*_STALLED_* made rocprofv3 hang for 25 minutes on this pool, round 3; every pass runs under `timeout`.)
scene=${1:-kitchen}; out=${2:-gpurun_out/pmc_diag}; shift; shift
mkdir -p $out; out=$(cd $out && pwd)
root=$(pwd)
groups=(
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"
"SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_IFETCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
"GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCC_REQ_sum"
"SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
)
cd /tmp; export TMPDIR=/tmp
k=0
for g in "${groups[@]}"; do
  timeout ${PMC_TIMEOUT:-100} rocprofv3 --pmc $g --output-format csv -d $out/g$k -- python3 $root/bench.py --scene $scene --pmc-child --steps 1 --warmup 0 --spp 32 --no-cpu-baseline --no-pmc --no-drop-in "$@" > $out/g$k.log 2>&1
  k=$((k+1))
done
cd $root
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
K = {"closest": "void skh::k_trace<false, false", "shadow": "void skh::k_trace<true, false", "shade": "void skh::k_shade<"}
res = {k: {} for k in K}
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        for name, p in K.items():
            if r["Kernel_Name"].startswith(p):
                per[(name, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for (name, c), d in per.items():
        res[name][c] = sum(d.values()) / len(d)
for name in K:
    print("==", name)
    for c in sorted(res[name]):
        print("   %-44s %.6g" % (c, res[name][c]))
PY
