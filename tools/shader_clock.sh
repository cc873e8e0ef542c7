# Effective shader clock of k_trace<closest> under rocprofv3: GRBM_GUI_ACTIVE / (End - Start) per dispatch (sum over the 8 XCDs).
# usage (GPU box): bash tools/shader_clock.sh <name> [DEFINE ...]
name=$1; shift
defs=""; for d in "$@"; do defs="$defs -D$d"; done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -std=c++17 $defs -o /tmp/lib_$name.so strelka_amd/csrc/strelka_hip.hip || exit 1
export SKH_LIB=/tmp/lib_$name.so
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clk_$name
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk_$name -- python3 $R/bench.py --steps 1 --warmup 0 --spp 32 --no-pmc --no-drop-in --no-cpu-baseline > /tmp/clk_$name.log 2>&1
cd $R
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob('/tmp/clk_$name/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('void skh::k_trace<false, false') and r['Counter_Name']=='GRBM_GUI_ACTIVE':
            rows.append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
for c,ns in rows:
    print('CLK $name cycles %.4g ns %d -> %.1f MHz, %.3f ms' % (c, ns, c/ns*1e3, ns/1e6))
PY
