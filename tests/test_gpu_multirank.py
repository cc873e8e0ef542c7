"""N > 1 launch path of bench.py on a 1-GPU box: two ranks share cuda:0 (SKH_DIST_BACKEND=gloo stages the tile gather
through the host, everything else -- tile assignment, per-rank rendering, gather, de-tiling scatter on the root, timing
reduction -- is the code the driver runs with RCCL on 8 GPUs).  The gathered image must equal the 1-rank image bit for bit."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--no-pmc", "--no-drop-in", "--scene", "cornell", "--width", "200", "--height", "136", "--spp", "5", "--depth", "4", "--steps", "1", "--warmup", "0",
        "--no-cpu-baseline"]


def _run(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_reproduce_the_single_rank_image():
    env = dict(os.environ, SKH_BENCH_CHECKSUM="1", SKH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, env)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533", "bench.py", "--gpus", "2"] + ARGS, env)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]
    assert two["image_crc32"] == one["image_crc32"]
    assert two["scaling"] == "strong" and two["value"] > 0
    # what the record says about the collective and the ranks (the 8-GPU run is read from exactly these fields)
    assert one["gather"] == "none" and one["rccl_nranks"] == 0 and "per_rank" not in one
    assert two["gather"] == "torch.distributed gather" and two["rccl_nranks"] == 0  # (gloo: two ranks on one GPU cannot form an RCCL communicator)
    pr = two["per_rank"]
    assert len(pr["ms_per_step"]["all"]) == 2 and pr["ms_per_step"]["min"] <= pr["ms_per_step"]["mean"] <= pr["ms_per_step"]["max"] <= two["ms_per_step"] * 1.05
    assert sum(pr["tiles"]) == ((200 + 31) // 32) * ((136 + 31) // 32) and 1.0 <= pr["rays_imbalance_max_over_mean"] < 1.5


@pytest.mark.gpu
def test_gpus_flag_alone_starts_the_ranks():
    """`python bench.py --gpus 2` with no launcher around it must itself become a 2-rank job (the driver's SCALE run may be
    started exactly like the N=1 line) and say so in the record."""
    env = dict(os.environ, SKH_BENCH_CHECKSUM="1", SKH_DIST_BACKEND="gloo", MASTER_PORT="29537")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, env)
    two = _run([sys.executable, "bench.py", "--gpus", "2"] + ARGS, env)
    assert two["n_gpus"] == 2 and two["config"]["world_size"] == 2
    assert two["image_crc32"] == one["image_crc32"]


@pytest.mark.gpu
def test_a_failed_communicator_is_recorded_with_its_error():
    """Two ranks on ONE GPU asking for the RCCL gather: ncclCommInitRank refuses (two ranks, one device).  Off the real path (gloo
    backend) all ranks agree on torch.distributed's gather and the record names the error; on the nccl backend the same situation
    ends the run with exit code 3 (bench.gather_fallback_allowed, tests/test_bench_launch.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SKH_DIST_BACKEND="gloo", SKH_GATHER="rccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29539", "bench.py", "--gpus", "2"] + ARGS
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    # (gloo carries the hand-shake here; the communicator attempt itself is the real skh_comm_init and fails the same way)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    rec = json.loads(lines[0])
    assert rec["gather"] == "torch.distributed gather" and rec["rccl_nranks"] == 0 and "gather_error" in rec


def _build_rccl_double(tmp_path):
    """tests/cpp/rccl_double.cpp -> a shared library with the ten nccl* symbols the loader binds, for N processes sharing one GPU"""
    out = str(tmp_path / "librccl_double.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", out,
                           os.path.join(ROOT, "tests", "cpp", "rccl_double.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lrt"])
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_the_rccl_branch_of_the_gather_runs_with_n_ranks(tmp_path, world):
    """skh_gather_tiles with commWorld > 1 -- ncclGroupStart, N-1 ncclRecv into the root's chunk offsets, one ncclSend per other rank from its
    zero-padded staging buffer, ncclGroupEnd -- executed for real, below the C ABI, by N ranks.  A 1-GPU box cannot form an RCCL communicator
    of more than one rank, so SKH_RCCL_LIB binds tests/cpp/rccl_double.cpp instead (same call sequence, buffers and sizes; shared-memory
    transport).  200 x 136 at tile 32 = 35 tiles: with 8 ranks three of them own 5 tiles and five own 4, so max_tiles padding is exercised.
    The gathered image must be the 1-rank image bit for bit and the record must name N ranks and say that the double was used."""
    lib = _build_rccl_double(tmp_path)
    env = dict(os.environ, SKH_BENCH_CHECKSUM="1", SKH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, env)
    env = dict(env, SKH_GATHER="rccl", SKH_RCCL_LIB=lib)
    many = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                 "--master-port", str(29541 + world), "bench.py", "--gpus", str(world)] + ARGS, env)
    assert many["n_gpus"] == world and many["rccl_nranks"] == world
    assert many["gather"].startswith("skh_gather_tiles") and "test double" in many["gather"] and "gather_error" not in many
    assert many["image_crc32"] == one["image_crc32"]
    assert many["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]
    tiles = many["per_rank"]["tiles"]
    assert sum(tiles) == 35 and (world != 8 or (max(tiles) == 5 and min(tiles) == 4))


def test_the_real_rccl_stays_the_default(monkeypatch):
    """SKH_RCCL_LIB is the only way to the double: without it the loader binds librccl.so.1, and bench.py still exits 3 when the real
    communicator cannot form on the nccl backend (tests/test_bench_launch.py covers gather_fallback_allowed)."""
    src = open(os.path.join(ROOT, "strelka_amd", "csrc", "strelka_hip.hip")).read()
    assert 'getenv("SKH_RCCL_LIB")' in src and '"librccl.so.1"' in src
    import bench

    monkeypatch.delenv("SKH_ALLOW_GATHER_FALLBACK", raising=False)
    assert not bench.gather_fallback_allowed("nccl", {})
