#!/bin/bash
# Calibrates FETCH_SIZE / TCC_EA0_RDREQ on known random record fetches: profiles/r03d_probe_calibration.txt.  usage (GPU box): bash tools/probe_calibrate.sh
# --pmc only (never combined with a trace domain: the counter_collection CSV carries Kernel_Name by itself); profiler output under /tmp.
R="$(cd "$(dirname "$0")/.." && pwd)" || exit 1
OUT=/tmp/skh_probe_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o p -f csv -- python3 "$R/tools/probe_calibrate.py" > "$OUT/fetch.log" 2>&1
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d "$OUT/rdreq" -o p -f csv -- python3 "$R/tools/probe_calibrate.py" > "$OUT/rdreq.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for d in ("fetch", "rdreq"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            if "probe" in k[0]: print(d, k, len(v), v[-1])
PY
tail -5 "$OUT/fetch.log"
