"""HdStrelka flattens Hydra instancing: one mesh per instance.  For such a scene the flattened world-space hierarchy
(option flatten = 1) duplicates nothing.  A/B of the two hierarchies on a kitchen stand-in WITHOUT mesh sharing (2000 unique meshes,
same 1.7 M triangles) and on the bench workload (153 meshes shared by 2022 instances).  usage (GPU box): python tools/flatten_ab.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from strelka_amd import capi, scene as S, scenes

W, H, SPP, DEPTH = 1920, 1080, 64, 4
for name, sc in (("unique meshes (2000 x 1)", scenes.kitchen_standin(n_meshes=2000, n_instances=2000, tri_lo=200, tri_hi=5000, target_tris=1.72e6)),
                 ("shared meshes (bench workload)", scenes.kitchen_standin())):
    arr = sc.arrays()
    for flatten in (0, 1):
        ctx = capi.Context(0)
        ctx.set_option("flatten", flatten)
        ctx.set_scene(arr); ctx.resize(W, H)
        p = S.frame_params(sc.getCamera(), W, H, subframe_index=0, samples_this_launch=1, spp_total=SPP, max_depth=DEPTH)
        ctx.set_option("count_traversal", 1); ctx.reset_stats(); ctx.render_subframes(p, SPP, None); c = ctx.stats(); ctx.set_option("count_traversal", 0)
        ctx.render_subframes(p, SPP, None)
        ctx.reset_stats(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2): ctx.render_subframes(p, SPP, None)
        dt = (time.perf_counter() - t0) / 2
        st = ctx.stats(); rays = (st["rays_radiance"] + st["rays_shadow"]) / 2
        print("FLATTEN", name, "| triangles %d instances %d | flatten %d: %.1f ms/frame, %.0f Mray/s, build %.0f ms, nodes/ray %.1f tris/ray %.2f inst/ray %.2f" % (
            len(arr["indices"]) // 3, len(arr["instances"]), flatten, dt * 1e3, rays / dt / 1e6, st["ms_build"], c["nodes_visited"][0] / c["rays_radiance"],
            c["prims_tested"][0] / c["rays_radiance"], c["instances_entered"][0] / c["rays_radiance"]), flush=True)
        ctx.close()
