"""Randomised call patterns through skh_render_subframe: one render() per sub-frame with camera moves, index jumps, frame ends, map()
copies and other API calls thrown in.  Every call must hand back exactly the image that one-pass-per-call rendering (speculate 0) gives,
whatever the cap and with or without a pass traced in flight (speculate_async).  usage (GPU box): python tools/fuzz_speculate.py <first> <last>"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

from strelka_amd import capi, scene as S, scenes  # noqa: E402

bad = 0
t0 = time.time()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rs = np.random.RandomState(seed)
    if seed % 5 == 4:
        sc = scenes.hair_standin(seed=seed, n_strands=200 + 30 * (seed % 7), n_cp=6 + seed % 4)  # the curve build of the kernels
    else:
        sc = scenes.kitchen_standin(seed=seed, n_meshes=6, n_instances=20 + seed % 17, tri_lo=50, tri_hi=600) if seed % 3 else scenes.cornell_box()
    arr = sc.arrays()
    w, h = int(rs.randint(24, 130)), int(rs.randint(16, 90))
    spp = int(rs.randint(3, 40))
    cams = [sc.getCamera()]
    for _ in range(3):
        c2 = S.Camera(fov=float(rs.uniform(35.0, 70.0)))
        c2.lookAt(tuple(rs.uniform(-2.5, 2.5, 3) + np.array([0.0, 1.5, 3.0])), tuple(rs.uniform(-0.5, 0.5, 3) + np.array([0.0, 1.0, 0.0])))
        cams.append(c2)
    # the script of calls: (camera, sub-frame index, extra action)
    calls, cam, idx = [], 0, 0
    for _ in range(int(rs.randint(20, 70))):
        r = rs.rand()
        if r < 0.08:
            cam, idx = int(rs.randint(0, len(cams))), 0  # camera move: the frame restarts
        elif r < 0.12:
            idx = min(spp - 1, idx + int(rs.randint(1, 4)))  # a skipped sub-frame
        elif idx >= spp:
            idx = 0  # frame finished: the caller starts it again
        calls.append((cam, idx, int(rs.randint(0, 12))))
        idx += 1
    depth = int(rs.randint(1, 6))
    img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    host = np.empty((h, w, 4), np.float32)

    def run(cap, asy):
        ctx = capi.Context(0)
        ctx.set_option("speculate", cap)
        ctx.set_option("speculate_async", asy)
        ctx.set_scene(arr)
        ctx.resize(w, h)
        outs = []
        for cam_, i, act in calls:
            ctx.render_subframe(S.frame_params(cams[cam_], w, h, subframe_index=i, spp_total=spp, max_depth=depth), img.data_ptr())
            if act == 5:
                ctx.tonemap(img.data_ptr(), w, h, 1 + act % 3, (1.0, 1.0, 1.0), 2.2)  # OptiXRender::render ends with tonemap() when enabled
            if act < 6:
                ctx.buffer_download(img.data_ptr(), host)  # map()
                outs.append(host.copy())
            elif act < 9:
                outs.append(img.cpu().numpy().copy())
            else:
                outs.append(ctx.read_accum())
            if act == 10:
                ctx.stats()
            if act == 11:
                ctx.set_option("overlap", int(rs.randint(0, 3)))  # a setter in the middle of a frame
        st = ctx.stats()
        ctx.close()
        return outs, st["rays_radiance"] + st["rays_shadow"]

    rs_state = rs.get_state()
    base, rays0 = run(0, 0)
    for cap, asy in ((int(rs.choice([2, 3, 8, 16, 64])), 1), (8, 0)):
        rs.set_state(rs_state)
        got, rays = run(cap, asy)
        ok = all(np.array_equal(a, b) for a, b in zip(base, got)) and abs(rays - rays0) <= 0.01 * rays0 + 64
        if not ok:
            bad += 1
            first = next((k for k, (a, b) in enumerate(zip(base, got)) if not np.array_equal(a, b)), None)
            print("seed", seed, "cap", cap, "async", asy, "first differing call", first, "of", len(calls), "rays", rays, "vs", rays0, flush=True)
print("fuzz_speculate done: %d seeds, %d failures, %.0f s" % (int(sys.argv[2]) - int(sys.argv[1]), bad, time.time() - t0))
