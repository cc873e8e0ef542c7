"""Strong-scaling emulation on ONE GPU: render the tile share of rank 3 of a 1-, 2-, 4-, 8-rank split of the 1080p frame and compare
with 1/N of the full frame.  usage (GPU box): python tools/tile_share.py [option=value ...]   (skh_set_option names,
e.g. overlap=0 for one-stream passes)"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import bench
from strelka_amd import build, capi, scene as S, tiles
build.build()
opts = [a.split("=") for a in sys.argv[1:]]
sc, arr, _ = bench.load_workload("kitchen"); cam = sc.getCamera()
W, H = 1920, 1080
base = None
for world in (1, 2, 4, 8):
    t = tiles.assign_tiles(W, H, 32, world, 3 % world)
    ctx = capi.Context(0)
    for k, v in opts: ctx.set_option(k, int(v))
    ctx.set_scene(arr); ctx.set_tiles(32, t if world > 1 else None); ctx.resize(W, H)
    p = S.frame_params(cam, W, H, subframe_index=0, samples_this_launch=1, spp_total=64, max_depth=4)
    ctx.render_subframes(p, 64, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): ctx.render_subframes(p, 64, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    if world == 1: base = dt
    print("SHARE", " ".join(sys.argv[1:]), "world", world, "ms/frame %.2f" % (dt * 1e3), "= %.1f %% of (full frame / world)" % (100 * base / world / dt), flush=True)
    ctx.close()
