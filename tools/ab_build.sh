# A/B builds of the HIP library (run on the GPU box): tools/ab_build.sh <name> [DEFINE[=value] ...] builds a variant with the given -D flags
# and prints its bench line's kernel times (STEPS, BENCH_ARGS, PROF_LINES in the environment; -DSKH_LANE_PROFILE adds the phase profile).
# Every A/B number in DESIGN.md section 4 came out of this script.
name=$1; shift
defs=""; for d in "$@"; do defs="$defs -D$d"; done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -std=c++17 $defs -o /tmp/lib_$name.so strelka_amd/csrc/strelka_hip.hip || exit 1
SKH_LIB=/tmp/lib_$name.so python bench.py --steps ${STEPS:-2} --warmup 1 --no-pmc --no-drop-in --no-cpu-baseline ${BENCH_ARGS:-} > /tmp/ab_$name.json 2> /tmp/ab_$name.err
python - <<PY
import json
d=json.load(open('/tmp/ab_$name.json')); k=d['kernel_ms_per_frame']; r=d['roofline']
print('AB $name', d['config']['bvh_build_ms'], d['value'], d['ms_per_step'], 'closest', k['ms_trace_closest'], 'shadow', k['ms_trace_shadow'], 'shade', k['ms_shade'], r['per_ray'], r['per_shadow_ray'])
PY
grep -E "lane-cycles" /tmp/ab_$name.err | tail -${PROF_LINES:-0}
