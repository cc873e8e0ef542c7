"""bench.py's launch logic (no GPU): --gpus N without a launcher spawns torch.distributed.run as a child process before torch
is imported; --gpus that disagrees with WORLD_SIZE is refused instead of silently recording a 1-rank run as N GPUs."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_builds_the_drivers_command(monkeypatch):
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "run", fake_run)
    rc = bench.maybe_self_launch(argparse.Namespace(gpus=4), ["--gpus", "4", "--steps", "3"])
    assert rc == 7  # the child's exit code is passed through
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # dmabuf IPC for RCCL's peer buffers, before any rank touches the GPU


def test_single_gpu_and_launched_ranks_do_not_relaunch(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.maybe_self_launch(argparse.Namespace(gpus=1), []) is None
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert bench.maybe_self_launch(argparse.Namespace(gpus=8), []) is None


def test_gpus_must_agree_with_world_size(monkeypatch, capsys):
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert bench.maybe_self_launch(argparse.Namespace(gpus=8), []) == 2
    assert "WORLD_SIZE=1" in capsys.readouterr().err


def test_bench_does_not_touch_torch_before_the_launch_decision():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("rc = maybe_self_launch")]
    assert "import torch" not in head.split("def main")[1]


def test_scene_cache_hands_every_process_the_same_scene(tmp_path, monkeypatch):
    """The counter child passes and the ranks of an N-GPU run load the .skscene one process wrote instead of regenerating the
    scene: arrays and camera must come back bit for bit (the N-rank image is compared with the 1-rank image by CRC)."""
    import numpy as np

    from strelka_amd import scene as S, scenes

    monkeypatch.setattr(bench, "scene_cache_path", lambda name: str(tmp_path / (name + ".skscene")))
    sc1, arr1, line1 = bench.load_workload("cornell")  # generates + writes
    assert os.path.exists(tmp_path / "cornell.skscene")
    sc2, arr2, line2 = bench.load_workload("cornell", make=False)  # what a rank != 0 does
    ref = scenes.cornell_box()
    refarr = ref.arrays()
    assert line1 == line2 and "cornell" in line1
    for k in ("vertices", "indices", "meshes", "instances", "lights", "materials"):
        assert np.array_equal(np.asarray(arr2[k]).view(np.uint8), np.asarray(refarr[k]).view(np.uint8)), k
    a = S.frame_params(sc2.getCamera(), 200, 136, subframe_index=3, spp_total=8, max_depth=4)
    b = S.frame_params(ref.getCamera(), 200, 136, subframe_index=3, spp_total=8, max_depth=4)
    assert a.tobytes() == b.tobytes()


def test_counter_arithmetic_and_profiler_detection(monkeypatch):
    f = bench.pmc_figures({"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0, "SQ_INSTS_VALU": 2e6, "SQ_INSTS_SALU": 5e5, "SQ_ACTIVE_INST_VALU": 1e6,
                           "SQ_THREAD_CYCLES_VALU": 3.2e7, "SQ_WAVE_CYCLES": 4e6, "SQ_WAIT_INST_ANY": 1e6}, 1000)
    assert f["hbm_bytes_per_launch"] == (2 * 1000.0 + 500.0) * 1024 and f["lanes_per_valu_inst"] == 32.0
    assert f["salu_per_valu"] == 0.25 and f["wait_inst_any_frac"] == 0.25 and f["rays_per_launch"] == 1000
    for k in list(os.environ):
        if k.startswith(("ROCP", "LD_PRELOAD")):
            monkeypatch.delenv(k)
    assert not bench.under_profiler()
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "librocprofiler-sdk-tool.so")
    assert bench.under_profiler()  # live_pmc refuses to nest profilers
    assert bench.live_pmc([], 64) is None
    # 36 + 20 + 64 nodes + 48 tris + 64 segs + 48 instances per radiance ray, 36 + 4 + ... per shadow ray
    assert bench.algorithmic_bytes(10, False, 20, 5, 0, 2) == 10 * 56 + 64 * 20 + 48 * 5 + 48 * 2
    assert bench.algorithmic_bytes(10, True, 0, 0, 1, 0) == 10 * 40 + 64


def test_a_real_multi_gpu_run_never_times_the_fallback_gather():
    assert not bench.gather_fallback_allowed("nccl", {})  # one rank per GPU: no communicator below the C ABI = exit code 3
    assert bench.gather_fallback_allowed("nccl", {"SKH_ALLOW_GATHER_FALLBACK": "1"})
    assert bench.gather_fallback_allowed("gloo", {})  # the 1-GPU tests: two ranks share cuda:0, RCCL refuses that by design


def test_committed_counter_files_are_kept_per_workload():
    """profiles/pmc_kernels.json belongs to the default workload; a profile of another scene must not replace it (round 3: a hair
    profile did), and a run that cannot collect counters only replays figures of ITS workload and resolution."""
    import json
    import os

    import bench

    assert os.path.basename(bench.pmc_file("kitchen", "1920x1080")) == "pmc_kernels.json"
    assert os.path.basename(bench.pmc_file("hair", "1920x1080")) == "pmc_kernels_hair_1920x1080.json"
    assert os.path.basename(bench.pmc_file("kitchen", "3840x2160")) == "pmc_kernels_kitchen_3840x2160.json"
    j = json.load(open(bench.pmc_file("kitchen", "1920x1080")))
    assert j["workload"].startswith("kitchen stand-in") and j["resolution"] == "1920x1080" and set(j["kernels"]) == {"closest", "shadow", "shade"}
    got = bench.committed_pmc(j["workload"], "1920x1080", "kitchen")
    assert got["replayed"] and "pmc_kernels.json" in got["source"]
    assert bench.committed_pmc(j["workload"], "3840x2160", "kitchen") is None
    assert bench.committed_pmc("some other scene", "1920x1080", "kitchen") is None


def test_roofline_block_states_both_fractions():
    """the counter fraction (`frac`, frac_kind counter_upper_bound) and SURVEY 8(d)'s algorithmic fraction must both be
    readable from the line without arithmetic; a model fraction above 1 is labelled, and the cache's share of the model's bytes is stated."""
    # round 3's closest-hit launch: 1649 B/ray x 130.8 M rays in 21.35 ms = 10.1 TB/s of model bytes; counters: 79.9 GB
    f = bench.roofline_fractions(1649 * 130.8e6, 79.9e9, 21.35)
    assert abs(f["algorithmic_frac"] - 1.263) < 2e-3 and f["model_exceeds_peak"] is True
    assert abs(f["l2_hit_share"] - (1 - 79.9e9 / (1649 * 130.8e6))) < 1e-3 and 0.6 < f["l2_hit_share"] < 0.65
    g = bench.roofline_fractions(4.0e9, None, 1.0)  # 4 TB/s of model bytes, no counters in this run
    assert g["algorithmic_frac"] == 0.5 and g["model_exceeds_peak"] is False and g["l2_hit_share"] is None
    assert bench.roofline_fractions(0, None, 0.0) == {"algorithmic_frac": None, "model_exceeds_peak": None, "l2_hit_share": None}
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"frac_kind": "counter_upper_bound"' in src and "str(min(spp, 32))" not in src  # counter children trace the timed run's pass size


def test_scene_cache_waiter_stops_when_the_writer_failed(tmp_path, monkeypatch):
    """a rank waiting for the node's writer must not sit out the whole time-out after the writer died."""
    import pytest

    monkeypatch.setattr(bench, "scene_cache_path", lambda name: str(tmp_path / (name + ".skscene")))

    def boom(scenes):
        raise ValueError("generator exploded")

    monkeypatch.setitem(bench.SCENE_RECIPES, "cornell", (boom, "x %d %d %d"))
    with pytest.raises(ValueError):
        bench.load_workload("cornell", make=True)
    assert os.path.exists(tmp_path / "cornell.skscene.failed")
    import time

    t0 = time.time()
    os.utime(tmp_path / "cornell.skscene.failed")  # (a marker written while we wait)
    with pytest.raises(RuntimeError, match="writer .* failed: ValueError: generator exploded"):
        bench.load_workload("cornell", make=False)
    assert time.time() - t0 < 5


def test_committed_round_profile_is_self_consistent():
    """The newest profiles/rNNx_bench.json and the rocprofv3 --kernel-trace --stats summary committed beside it describe the same code: the dominant
    kernel's average launch in the bench line (hipEvents inside bench.py) agrees with the profiler's (within 5 %), the fractions follow from the
    line's own bytes / time / peak, and the per-kernel times add up to the step."""
    import csv
    import glob
    import json
    import re

    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    tags = sorted({re.match(r"(r\d+[a-z]?)_bench\.json$", os.path.basename(f)).group(1) for f in glob.glob(os.path.join(prof, "r*_bench.json"))
                   if re.match(r"r\d+[a-z]?_bench\.json$", os.path.basename(f))})
    tag = tags[-1]
    line = json.load(open(os.path.join(prof, tag + "_bench.json")))
    roof = line["roofline"]
    rows = list(csv.DictReader(open(os.path.join(prof, tag + "_kernel_stats.csv"))))
    closest = [r for r in rows if "k_trace<false, false" in r["Name"]]
    assert closest, "closest-hit kernel missing from the committed kernel stats"
    avg_ms = float(closest[0]["AverageNs"]) / 1e6
    assert abs(avg_ms - roof["avg_launch_ms"]) <= 0.05 * roof["avg_launch_ms"], (avg_ms, roof["avg_launch_ms"])
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["unit"] == "GB/s" and roof["bound"] == "hbm"
    assert abs(roof["achieved"] - roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * roof["achieved"]
    k = line["kernel_ms_per_frame"]
    assert abs(sum(k.values()) - line["ms_per_step"]) <= 0.02 * line["ms_per_step"]
    assert abs(line["value"] - line["config"]["rays_per_frame"] / line["ms_per_step"] / 1e3) <= 0.01 * line["value"]


def test_limiter_is_derived_from_the_lines_own_fractions():
    """roofline.limiter names the largest of the HBM-byte, random-line-rate and VALU-issue fractions when it is above 0.6 and says
    "latency / divergence" otherwise (round 4 printed a string literal whatever the counters said); frac_of_measured_copy prices the
    counter bytes against this box's measured copy rate instead of the 8 TB/s data-sheet peak."""
    blk = {"frac": 0.53, "achieved": 4275.9, "l2_miss_lines": {"frac_of_measured_random_line_rate": 0.56},
           "valu": {"frac_valu_issue": 0.54, "lanes_per_valu_inst": 37.9}}
    lim = bench.derive_limiter(blk, None)
    assert lim["name"].startswith("latency / divergence") and "random_line_rate 0.56" in lim["name"] and "37.9 of 64 lanes" in lim["name"]
    assert lim["fractions"] == {"hbm_bytes": 0.53, "random_line_rate": 0.56, "valu_issue": 0.54}
    assert bench.derive_limiter(dict(blk, frac=0.66), None)["name"] == "hbm_bytes"
    assert bench.derive_limiter(dict(blk, valu={"frac_valu_issue": 0.9}), None)["name"] == "valu_issue"
    assert bench.derive_limiter({}, None)["name"].startswith("unknown")
    assert bench.frac_of_copy(blk, {"stream_copy_GBps": 4386.5}) == round(4275.9 / 4386.5, 4)
    assert bench.frac_of_copy(blk, None) is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"limiter": "' not in src  # no literal limiter left
    assert 'extra["hair"]' in src  # C5 is a leg of the default line


def test_measured_clock_from_the_counter_pass(tmp_path):
    """Round 6: the VALU-issue roof is priced against the clock the launches really ran at.  `_pmc_per_launch` turns the
    GRBM_GUI_ACTIVE pass into GHz -- busy cycles / dispatch duration, per engine instance whether rocprofv3 writes one row per XCD or one row with
    the sum of the eight -- and `pmc_figures` hands it on; the record carries `value_drop_in`, `interactive` and a `config.step` that says what the
    headline's step is."""
    head = "Kernel_Name,Counter_Name,Counter_Value,Dispatch_Id,Start_Timestamp,End_Timestamp\n"
    k = '"void skh::k_trace<false, false, false, true>(skh::DevScene)"'
    d = tmp_path / "perxcd"
    d.mkdir()
    # one row per XCD: 8 rows x 2.3e7 busy cycles over 10 ms -> 2.3 GHz; two dispatches
    (d / "1_counter_collection.csv").write_text(head + "".join(f"{k},GRBM_GUI_ACTIVE,{2.3e7 if disp == 1 else 2.2e7},{disp},1000,{1000 + 10_000_000}\n"
                                                                 for disp in (1, 2) for _ in range(8)))
    got = bench._pmc_per_launch(str(d), bench.CLOSEST)
    assert abs(got["__clock_ghz"] - 2.25) < 1e-6
    d2 = tmp_path / "summed"
    d2.mkdir()
    (d2 / "1_counter_collection.csv").write_text(head + f"{k},GRBM_GUI_ACTIVE,{8 * 2.31e7},7,0,10000000\n")
    assert abs(bench._pmc_per_launch(str(d2), bench.CLOSEST)["__clock_ghz"] - 2.31) < 1e-6
    fig = bench.pmc_figures({"__clock_ghz": 2.31, "SQ_INSTS_VALU": 1e9}, 1000)
    assert fig["clock_ghz_measured"] == 2.31
    assert ("GRBM_GUI_ACTIVE",) in bench.PMC_PASSES
    src = open(os.path.join(ROOT, "bench.py")).read()
    for needle in ('out["value_drop_in"]', 'out["interactive"]', "one skh_render_subframes(", "frac_valu_issue_at_max_clock", 'extra["hair_multi"]'):
        assert needle in src, needle
