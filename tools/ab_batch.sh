# A/B of the batch-scaling probe (GPU box): tools/ab_batch.sh <name> "<probe args>" [DEFINE[=value] ...]
name=$1; pargs=$2; shift; shift
defs=""; for d in "$@"; do defs="$defs -D$d"; done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -std=c++17 $defs -o /tmp/lib_$name.so strelka_amd/csrc/strelka_hip.hip || exit 1
SKH_LIB=/tmp/lib_$name.so timeout 300 python tools/batch_scaling_probe.py kitchen $pargs 2>&1 | grep "n=64\|n= 1:\|FIT" | sed "s/^/AB $name: /"
