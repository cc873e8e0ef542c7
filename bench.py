#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

metric : Mray/s (+ ms/frame) at 1920x1080, 4 bounces, 64 spp, kitchen-class scene, full BSDF set
step   : one FRAME = 64 sub-frame launches of 1 spp (reference default spp=1: src/hdRunner/main.cpp:454,514),
         excluding scene upload, BVH build and D2H read-back; scene + BVH are resident in HBM
         when the timed region starts.
rays   : radiance segments + shadow rays ACTUALLY traced (device queue lengths), not W*H*spp*depth.
N > 1  : one process per GPU (torch.distributed, backend nccl = RCCL); the frame's pixel tiles are dealt round-robin
         to the ranks (no data-path collective), one gather of the tile accumulators to rank 0 per frame.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def algorithmic_bytes(rays, shadow, nodes, prims, segs, insts):
    """bytes/ray = 36 + {20 | 4} + 64 N_node + 48 N_tri + 64 N_seg + 48 N_inst."""
    return rays * (36 + (4 if shadow else 20)) + 64 * nodes + 48 * prims + 64 * segs + 48 * insts


def cpu_baseline(arr, cam, width, height, spp_total, depth, budget_s=12.0):
    """The CPU oracle (kind "port": the reference has no CPU path, include/render/render.h:9-14 + render.cpp:10-35)
    timed on this box's host cores over a bounded sample: sub-frame 0, rows added until ~budget_s of work."""
    from strelka_amd import scene as S
    from tests import orklib

    # threads = the CPUs this process may really use: affinity mask AND the cgroup CPU quota (a container that sees 256
    # hardware threads but is throttled to 16 CPUs runs 128 OpenMP threads slower than 16)
    usable = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    orklib.load().ork_set_num_threads(usable)
    o = orklib.new_context()
    t0 = time.time()
    o.set_scene(arr)
    build_s = time.time() - t0
    o.resize(width, height)
    p = S.frame_params(cam, width, height, subframe_index=0, spp_total=spp_total, max_depth=depth)
    rows, y, t_used = 8, 0, 0.0
    while y < height and t_used < budget_s:
        y1 = min(height, y + rows)
        t0 = time.time()
        o.render_subframe(p, rows=(y, y1))
        t_used += time.time() - t0
        y = y1
        rows = min(rows * 2, 128)
    st = o.stats()
    rays = st["rays_radiance"] + st["rays_shadow"]
    return {"value": round(rays / t_used / 1e6, 4), "unit": "Mray/s", "cores": int(orklib.load().ork_num_threads()),
            "kind": "port",
            "sample": f"sub-frame 0 (1 spp), image rows 0..{y} of {height} at {width}x{height}, depth {depth}: "
                      f"{rays} rays in {t_used:.2f} s; oracle BVH build {build_s:.2f} s not included"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--scene", default="kitchen", help="kitchen | cornell | hair | path to a .skscene dump or a .gltf file")
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to skh_set_option")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SKH_DIST_BACKEND", "nccl")  # "gloo": several ranks on ONE GPU (tests on a 1-GPU box)
        if backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)

    from strelka_amd import build, capi, scene as S, scenes, tiles

    build.build()
    if args.scene == "kitchen":
        sc = scenes.kitchen_standin()
        workload = ("kitchen stand-in (SURVEY 8d C3): %d unique triangles, %d instances of %d meshes, "
                    "4 rect + 1 distant light, 60/25/10/5 %% diffuse/glossy/metal/glass")
    elif args.scene == "hair":
        sc = scenes.hair_standin()
        workload = "hair stand-in (SURVEY 8d C5): %d scalp triangles, %d instances of %d meshes + 100 k strands"
    elif args.scene == "cornell":
        sc = scenes.cornell_box()
        workload = "cornell box (C2): %d triangles, %d instances of %d meshes"
    else:
        # a flat dump of a real bake (strelka_amd/scene_io.py) or a glTF file
        from strelka_amd import scene_io

        if args.scene.endswith((".gltf", ".glb")):
            from strelka_amd import gltf

            sc = gltf.load_gltf(args.scene)
        else:
            sc = scene_io.load_scene(args.scene)
        workload = os.path.basename(args.scene) + ": %d triangles, %d instances of %d meshes"
    arr = sc.arrays()
    workload = workload % (len(arr["indices"]) // 3, len(arr["instances"]), len(arr["meshes"]))
    cam = sc.getCamera()
    W, H = args.width, args.height

    ctx = capi.Context(local_rank)
    if args.waves_per_cu:
        ctx.set_option("waves_per_cu", args.waves_per_cu)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    ctx.set_scene(arr)
    my_tiles = tiles.assign_tiles(W, H, args.tile, world, rank)
    ctx.set_tiles(args.tile, my_tiles if world > 1 else None)
    ctx.resize(W, H)
    build_ms = ctx.stats()["ms_build"]
    params = S.frame_params(cam, W, H, subframe_index=0, samples_this_launch=1, spp_total=args.spp, max_depth=args.depth)

    max_tiles = tiles.max_tiles_per_rank(W, H, args.tile, world)
    tile_buf = torch.zeros((max_tiles, args.tile * args.tile, 4), dtype=torch.float32, device=dev)
    image = torch.zeros((H, W, 4), dtype=torch.float32, device=dev) if rank == 0 else None
    # root: one receive buffer for all ranks' tiles and ONE de-tiling launch per frame; a rank with fewer tiles than
    # max_tiles sends padding, which gets an origin outside the image and is dropped by the scatter kernel
    all_tiles = torch.zeros((world, max_tiles, args.tile * args.tile, 4), dtype=torch.float32, device=dev) if rank == 0 and world > 1 else None
    all_xy = None
    if rank == 0 and world > 1:
        all_xy = np.full((world, max_tiles, 2), max(W, H), np.uint32)
        for r in range(world):
            t = tiles.assign_tiles(W, H, args.tile, world, r)
            all_xy[r, :len(t)] = t
        all_xy = np.ascontiguousarray(all_xy.reshape(-1, 2))

    def frame():
        ctx.render_subframes(params, args.spp, None)  # all sub-frames of the frame, one device sync at the end
        if world > 1:
            ctx.copy_accum_tiles(tile_buf.data_ptr())
            tiles.gather_tiles(tile_buf, world, rank, dist, out=all_tiles)
            if rank == 0:
                ctx.scatter_tiles(all_tiles.data_ptr(), all_xy, args.tile, image.data_ptr(), W, H)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed counting pass: exact traversal counters for the algorithmic-bytes figure (identical every frame,
    #      the renderer is deterministic) ----
    ctx.set_option("count_traversal", 1)
    ctx.reset_stats()
    ctx.render_subframes(params, args.spp, None)
    cst = ctx.stats()
    ctx.set_option("count_traversal", 0)
    for _ in range(args.warmup):
        frame()
    ctx.set_option("timing", 1)
    ctx.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    barrier()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    rays_local = st["rays_radiance"] + st["rays_shadow"]
    if world > 1:
        rdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        rr = torch.tensor([float(rays_local)], dtype=torch.float64, device=rdev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays_total = float(rr.item())
    else:
        rays_total = float(rays_local)

    if rank == 0:
        K = max(1, args.steps)
        # roofline of the dominant kernel: k_trace<closest>
        bytes_closest = algorithmic_bytes(cst["rays_radiance"], False, cst["nodes_visited"][0], cst["prims_tested"][0],
                                          cst["segs_tested"][0], cst["instances_entered"][0])
        launches = max(1, st["launches_trace_closest"])
        avg_ms = st["ms_trace_closest"] / launches
        bytes_per_launch = bytes_closest * K / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_k_trace_closest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # PMC bytes per launch, measured on launches of `rays_per_launch` rays of the same workload; quoted per
                # launch of THIS run (same figure when the launch sizes agree, which they do for the default config)
                if tj.get("workload", workload) == workload and tj.get("resolution", f"{W}x{H}") == f"{W}x{H}":
                    rpl = cst["rays_radiance"] * K / launches
                    traffic = int(tj["hbm_bytes_per_launch"] * (rpl / tj["rays_per_launch"] if tj.get("rays_per_launch") else 1.0))
            except Exception:
                traffic = None
        out = {
            "metric": "Mray/s", "value": round(rays_total / dt / 1e6, 3), "unit": "Mray/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "resolution": f"{W}x{H}", "bounces": args.depth, "spp": args.spp,
                       "step": f"one frame = {args.spp} sub-frames of 1 spp", "tile": args.tile,
                       "parallelism": f"pixel tiles round-robin over {world} GPU(s), 1 RCCL gather/frame" if world > 1
                       else "single GPU", "rays_per_frame": int(rays_total / K), "bvh_build_ms": round(build_ms, 2)},
            "kernel_ms_per_frame": {k: round(st[k] / K, 3) for k in ("ms_trace_closest", "ms_trace_shadow", "ms_shade",
                                                                     "ms_raygen", "ms_accumulate", "ms_sort")},
            "roofline": {"kernel": "k_trace<closest>", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "rays_per_launch": int(cst["rays_radiance"] * K / launches),
                         "per_ray": {"nodes": round(cst["nodes_visited"][0] / max(1, cst["rays_radiance"]), 2),
                                     "tris": round(cst["prims_tested"][0] / max(1, cst["rays_radiance"]), 2),
                                     "instances": round(cst["instances_entered"][0] / max(1, cst["rays_radiance"]), 2)}},
        }
        if os.environ.get("SKH_BENCH_CHECKSUM"):
            # CRC of the final accumulation image (tests: a tile-sharded N-rank run must reproduce the 1-rank image exactly)
            import zlib

            img = image.cpu().numpy() if world > 1 else ctx.read_accum()
            out["image_crc32"] = zlib.crc32(np.ascontiguousarray(img[..., :3]).tobytes())
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(arr, cam, W, H, args.spp, args.depth, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
