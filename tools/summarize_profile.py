#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile.sh) into the tracked artefacts under profiles/:
  profiles/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary (per-kernel time)
  profiles/<tag>_pmc_hbm.csv               per-kernel FETCH_SIZE / WRITE_SIZE averages from the two --pmc passes
  profiles/<tag>_pmc_instructions.csv / _pmc_sq_utilisation.csv   SQ counters of the same passes
  profiles/<tag>_bench.json, profiles/pmc_kernels[_<scene>_<res>].json   the bench line of that run and the counter figures bench.py replays
                                                                     (tagged as replayed) when it cannot run rocprofv3 itself
Units/corrections follow /opt/skills/guides (MI355X_MICROARCH.md "HBM", cdna_hip_programming.md section 7):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) reads, so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The 2x is calibrated for coalesced streams; a BVH walk issues
16 B/lane loads to scattered 64 B nodes, so treat the absolute as +-2x and use it for ratios between builds.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace("void ", "").replace("skh::", "")
    return n.split("(")[0]


def pmc(dirpath, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                a = acc[short(row["Kernel_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    logs = os.path.join(src, "bench_trace.log")
    if os.path.exists(logs):
        lines = [l for l in open(logs) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(lines[-1])
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
    # the counter figures bench.py replays: one file per workload (pmc_kernels.json = the default one), only the one this run wrote

    for f in glob.glob(os.path.join(src, "pmc_kernels*.json")):
        try:
            if json.load(open(f)).get("tag") == tag:
                shutil.copy(f, os.path.join(dst, os.path.basename(f)))
        except (OSError, ValueError):
            pass
    # the counter passes bench.py ran itself (--pmc-keep): pass0 FETCH_SIZE, pass1 WRITE_SIZE, pass2 SQ counters
    fetch = pmc(os.path.join(src, "pmc", "pass0"), "FETCH_SIZE")
    write = pmc(os.path.join(src, "pmc", "pass1"), "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        f = fetch[k][0] / max(1, fetch[k][1]) if k in fetch else 0.0
        w = write[k][0] / max(1, write[k][1]) if k in write else 0.0
        rows.append((k, fetch[k][1] if k in fetch else 0, f, w, (2 * f + w) * 1024))
    with open(os.path.join(dst, f"{tag}_pmc_hbm.csv"), "w") as fo:
        fo.write("kernel,launches,avg_FETCH_SIZE_KiB,avg_WRITE_SIZE_KiB,hbm_bytes_per_launch=(2*FETCH+WRITE)*1024\n")
        for r in rows:
            fo.write("%s,%d,%.3f,%.3f,%.0f\n" % r)
    print(open(os.path.join(dst, f"{tag}_pmc_hbm.csv")).read())
    sq_summaries(src, dst, tag)


def sq_summaries(src, dst, tag):
    """optional passes (tools/profile.sh does not run them by default): <src>/pmc_inst with SQ_WAVES SQ_INSTS_*, and
    <src>/pmc_util with SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"""
    def collect(sub, key):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == key:
                    cnt[k] += 1
        return acc, cnt

    acc, cnt = collect(os.path.join("pmc", "pass2"), "SQ_WAVES")
    if acc:
        with open(os.path.join(dst, f"{tag}_pmc_instructions.csv"), "w") as fo:
            fo.write("kernel,launches,waves_per_launch,VALU_per_wave,SALU_per_wave,SALU_per_VALU\n")
            for k, a in sorted(acc.items()):
                w = a["SQ_WAVES"]
                if w and cnt[k]:
                    fo.write('"%s",%d,%.0f,%.0f,%.0f,%.3f\n' % (k, cnt[k], w / cnt[k], a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_SALU"] / w,
                                                              a["SQ_INSTS_SALU"] / max(1.0, a["SQ_INSTS_VALU"])))
    acc, cnt = collect(os.path.join("pmc", "pass2"), "SQ_WAVE_CYCLES")
    if acc:
        with open(os.path.join(dst, f"{tag}_pmc_sq_utilisation.csv"), "w") as fo:
            fo.write("kernel,launches,SQ_WAVE_CYCLES,SQ_ACTIVE_INST_VALU,SQ_THREAD_CYCLES_VALU,SQ_WAIT_INST_ANY,lanes_per_valu_inst,wait_inst_any_frac\n")
            for k, a in sorted(acc.items()):
                n = cnt[k]
                if n and a["SQ_ACTIVE_INST_VALU"]:
                    fo.write('"%s",%d,%.4g,%.4g,%.4g,%.4g,%.1f,%.3f\n' % (k, n, a["SQ_WAVE_CYCLES"] / n, a["SQ_ACTIVE_INST_VALU"] / n,
                                                                        a["SQ_THREAD_CYCLES_VALU"] / n, a["SQ_WAIT_INST_ANY"] / n,
                                                                        a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"],
                                                                        a["SQ_WAIT_INST_ANY"] / max(1.0, a["SQ_WAVE_CYCLES"])))


if __name__ == "__main__":
    main()
