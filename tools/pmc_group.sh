# One rocprofv3 --pmc pass over a one-frame bench run for an arbitrary counter group, per-kernel averages of the trace / shade kernels.
# usage (GPU box): bash tools/pmc_group.sh TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE      (150 s timeout: some TA / TCP groups stall rocprofv3)
R=$PWD
cd /tmp && export TMPDIR=/tmp
grp="$*"
rm -rf /tmp/pmc_ta
timeout 150 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_ta -- python3 $R/bench.py --pmc-child --steps 1 --warmup 0 --spp 32 --no-cpu-baseline --no-pmc --no-drop-in > /tmp/pmc_ta.log 2>&1
echo "TA group [$grp] rc=$?"
python3 - <<PY
import csv, glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/pmc_ta/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if n.startswith('void skh::k_trace<false, false') or n.startswith('void skh::k_trace<true, false') or n.startswith('void skh::k_shade'):
            key = 'closest' if 'k_trace<false' in n else ('shadow' if 'k_trace<true' in n else 'shade')
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print('TA', k, {c: '%.4g' % (sum(v)/len(v)) for c,v in acc[k].items()})
PY
