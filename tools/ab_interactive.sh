# usage: ab_inter.sh name "probe args" [DEFINES...]
name=$1; pargs=$2; shift; shift
defs=""; for d in "$@"; do defs="$defs -D$d"; done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -std=c++17 $defs -o /tmp/lib_$name.so strelka_amd/csrc/strelka_hip.hip || exit 1
SKH_LIB=/tmp/lib_$name.so timeout 300 python tools/interactive_probe.py kitchen $pargs 2>&1 | grep LOOP | sed "s/^/AB $name: /"
