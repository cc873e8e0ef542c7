# Calibrates FETCH_SIZE / TCC_EA0_RDREQ on known random record fetches: profiles/r03d_probe_calibration.txt.  usage (GPU box): bash tools/probe_calibrate.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/probe_pmc/fetch -o p -f csv -- python3 tools/probe_calibrate.py > gpurun_out/probe_pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d gpurun_out/probe_pmc/rdreq -o p -f csv -- python3 tools/probe_calibrate.py > gpurun_out/probe_pmc_rdreq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("fetch", "rdreq"):
    for f in glob.glob("gpurun_out/probe_pmc/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            if "probe" in k[0]: print(d, k, len(v), v[-1])
PY
tail -5 gpurun_out/probe_pmc_fetch.log
