"""Randomised render parity: small procedural scenes (all stand-in kinds, random extra instance transforms, optional textures),
random resolution / spp / depth / frame parameters; the GPU image must EQUAL the oracle's (bit for bit since round 5) and the
two sides must trace (almost) the same number of rays.  usage: python tools/fuzz_render.py <first seed> <last seed>  (GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from strelka_amd import capi, scene as S, scenes  # noqa: E402
from tests import orklib  # noqa: E402
from tests.test_gpu_parity import thick_curves  # noqa: E402
from tests.test_textures import checker, textured_scene  # noqa: E402

bad = 0
t0 = time.time()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rs = np.random.RandomState(seed)
    kind = seed % 7
    if os.environ.get("SKH_FUZZ_KIND") and kind != int(os.environ["SKH_FUZZ_KIND"]):
        continue  # (one family only: chasing an event)
    if kind == 5:
        sc = scenes.light_zoo(seed=seed, with_rect=bool(seed % 2))  # sphere + disk lights (+ rect + distant)
    elif kind == 6:
        sc = scenes.material_probe(["diffuse", "glossy", "metal", "glass", "frosted"][(seed // 7) % 5], seed=seed)
    elif kind == 0:
        sc = scenes.kitchen_standin(seed=seed, n_meshes=8 + seed % 7, n_instances=30 + 11 * (seed % 5), tri_lo=50, tri_hi=2000)
    elif kind == 1:
        npr = 1 + (seed // 7) % 4  # (round 6: 1..4 curve prims, some moved)
        sc = scenes.hair_standin(seed=seed, n_strands=300 + 40 * (seed % 9), n_cp=6 + seed % 5, n_prims=npr, prim_offset=0.01 if (seed // 28) % 2 else 0.0, n_moved=(seed // 56) % (npr + 1), shared_xform=bool((seed // 112) % 2))
    elif kind == 2:
        sc = thick_curves(seed=seed, n_strands=20 + seed % 30, n_cp=5 + seed % 6)
        sc.createLight({"type": 0, "useXform": False, "position": (0.0, 3.0, 1.0), "orientation": (-70.0, 0.0, 0.0), "width": 2.0, "height": 2.0,
                        "color": (1.0, 1.0, 1.0), "intensity": 20.0})
    elif kind == 3:
        sc = scenes.cornell_box()
    else:
        bumps = rs.randint(90, 170, (8, 8, 4)).astype(np.uint8)
        bumps[..., 2] = 255
        sc = textured_scene(base_tex=checker(4 + seed % 5), normal_tex=bumps if seed % 2 else None)
    arr = dict(sc.arrays())
    w, h = int(rs.randint(17, 120)), int(rs.randint(11, 90))
    spp, depth = int(rs.randint(1, 6)), int(rs.randint(1, 7))
    kw = {"rect_light_sampling_method": int(rs.randint(0, 2)), "max_depth": depth}
    import zlib as _z

    crc = lambda: {k: _z.crc32(np.ascontiguousarray(v).tobytes()) for k, v in arr.items() if isinstance(v, np.ndarray)}
    o = orklib.new_context()
    crc_o = crc()  # the arrays as the oracle receives them
    o.set_scene(arr)
    o.resize(w, h)
    ctx = capi.Context(0)
    gpu_opts = {"subframe_batch": int(rs.choice([0, 1, 2])), "overlap": 1, "speculate": 8}
    ctx.set_option("subframe_batch", gpu_opts["subframe_batch"])
    ctx.set_option("direct_records", (-1, 0, 1)[seed % 3])  # which word a baked triangle's hit carries: the image does not depend on it
    ctx.set_option("compact_hits", 0 if seed % 5 == 4 else 1)
    ctx.set_option("curve_merge", (seed // 9) % 2)
    ctx.set_option("tail_split", 2 * ((seed // 4) % 2))  # the SPLIT build of the world-only triangle kernels for every pass, or never: the image does not depend on it
    crc_g = crc()  # ... and as the GPU receives them
    ctx.set_scene(arr)
    ctx.set_tiles(int(rs.choice([8, 16, 32, 64])), None)
    ctx.resize(w, h)
    p0 = S.frame_params(sc.getCamera(), w, h, subframe_index=0, spp_total=spp, **kw)
    for i in range(spp):
        o.render_subframe(S.frame_params(sc.getCamera(), w, h, subframe_index=i, spp_total=spp, **kw))
    ctx.render_subframes(p0, spp, None)
    want, got = o.read_accum()[..., :3].astype(np.float64), ctx.read_accum()[..., :3].astype(np.float64)
    so_, sg_ = o.stats(), ctx.stats()
    ro, rg = so_["rays_radiance"], sg_["rays_radiance"]
    ctx.close()
    # the bar of tests/test_gpu_parity.py::_image_equal: EQUAL, and the same radiance-ray count.  (Rounds 2-4 allowed up to 2.5 flipped paths per
    # frame -- 4.5 with the spherical-rectangle sampler, 1e-3 of the pixels with the hair BSDF -- for the few-ulp differences between glibc and the
    # ROCm device library; both sides compile strelka_amd/csrc/skh_libm.h now.)
    dev = np.abs(got - want).max(-1)
    frac = float((dev > 0).mean())
    l2 = np.sqrt(((got - want) ** 2).sum()) / max(np.sqrt((want ** 2).sum()), 1e-12)
    ok = np.isfinite(got).all() and frac == 0.0 and int(ro) == int(rg)
    if not ok:
        bad += 1
        # render the GPU side once more: a different answer now means a race / an uninitialised read, the same answer a real disagreement
        ctx = capi.Context(0)
        ctx.set_scene(arr)
        ctx.resize(w, h)
        ctx.render_subframes(p0, spp, None)
        again = ctx.read_accum()[..., :3].astype(np.float64)
        rg2 = ctx.stats()["rays_radiance"]
        ctx.close()
        # ... and the oracle side once more (round 2: two unexplained one-off events, one per side -- seed 5133 the GPU's first answer, seed 8283 the
        # oracle's; every later run of either side agreed with the other side's number)
        o2 = orklib.new_context()
        o2.set_scene(arr)
        o2.resize(w, h)
        for i in range(spp):
            o2.render_subframe(S.frame_params(sc.getCamera(), w, h, subframe_index=i, spp_total=spp, **kw))
        want2 = o2.read_accum()[..., :3].astype(np.float64)
        ro2 = o2.stats()["rays_radiance"]
        # everything needed to name the cause: the scene as both sides received it, checksums of every array at the moment each side
        # took it, both stats blocks, the stream settings, a GPU render on ONE stream, and the worst pixel's path -- the oracle's rays,
        # one by one, re-traced through the GPU's and the oracle's raw query entry (a flipped path shows up as the first ray whose
        # hit differs, or as none: then the two sides shaded the same hits differently)
        import json
        import zlib

        from strelka_amd import scene_io

        outdir = os.path.join("gpurun_out", "fuzz_fail_%d" % seed)
        os.makedirs(outdir, exist_ok=True)
        scene_io.save_scene(os.path.join(outdir, "scene.skscene"), arr, sc.getCamera())
        crc_now = {k: zlib.crc32(np.ascontiguousarray(v).tobytes()) for k, v in arr.items() if isinstance(v, np.ndarray)}
        ctx1 = capi.Context(0)
        ctx1.set_option("overlap", 0)
        ctx1.set_option("speculate", 0)
        ctx1.set_scene(arr)
        ctx1.resize(w, h)
        ctx1.render_subframes(p0, spp, None)
        one_stream = ctx1.read_accum()[..., :3].astype(np.float64)
        wy, wx = np.unravel_index(int(np.argmax(dev)), dev.shape)
        trail = []
        for i in range(spp):
            pp = S.frame_params(sc.getCamera(), w, h, subframe_index=i, spp_total=spp, **kw)
            path, rad = o2.debug_path(pp, int(wx), int(wy), i)
            rr = np.zeros(len(path), S.RAY)
            rr["origin"], rr["tmin"], rr["dir"], rr["tmax"] = path[:, 1:4], path[:, 4], path[:, 5:8], path[:, 8]
            for k_, row in enumerate(path):
                gh, oh = ctx1.trace(rr[k_:k_ + 1], int(row[0]))[0], o2.trace(rr[k_:k_ + 1], int(row[0]))[0]
                trail.append({"sample": i, "kind": "shadow" if row[0] else "radiance", "oracle_in_path": [float(row[9]), int(row[10]), int(row[11])],
                              "gpu_trace": [float(gh["t"]), int(gh["instance_id"]), int(gh["prim_id"])],
                              "oracle_trace": [float(oh["t"]), int(oh["instance_id"]), int(oh["prim_id"])]})
        st1 = ctx1.stats()
        ctx1.close()
        json.dump({"seed": seed, "kind": kind, "w": w, "h": h, "spp": spp, "depth": depth, "l2": l2, "frac": frac,
                   "crc32_at_oracle_upload": crc_o, "crc32_at_gpu_upload": crc_g, "crc32_now": crc_now,
                   "inputs_identical": crc_o == crc_g == crc_now, "stats_oracle": so_, "stats_gpu": sg_, "stats_gpu_one_stream": st1,
                   "gpu_options": gpu_opts, "one_stream_render_equals_first": bool(np.array_equal(one_stream, got)),
                   "worst_pixel": [int(wx), int(wy)], "worst_pixel_gpu": got[wy, wx].tolist(), "worst_pixel_oracle": want[wy, wx].tolist(),
                   "worst_pixel_path": trail}, open(os.path.join(outdir, "report.json"), "w"), indent=1)
        print("seed", seed, "kind", kind, w, h, spp, depth, "L2 %.3g frac %.3g rays oracle %d vs GPU %d" % (l2, frac, ro, rg),
              "| inputs identical: %s | one-stream GPU render %s the first | report: %s" % (crc_o == crc_g == crc_now, "equals" if np.array_equal(one_stream, got) else "DIFFERS from", outdir),
              "| second GPU render: rays %d, %s the first" % (rg2, "equals" if np.array_equal(again, got) else "DIFFERS from"),
              "| second oracle render: rays %d, %s the first" % (ro2, "equals" if np.array_equal(want2, want) else "DIFFERS from"), flush=True)
print("fuzz done: %d seeds, %d failures, %.0f s" % (int(sys.argv[2]) - int(sys.argv[1]), bad, time.time() - t0))
