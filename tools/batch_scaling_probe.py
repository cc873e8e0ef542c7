"""How the three kernels' time grows with the paths per launch: skh_render_subframes of n = 1 ... 64 sub-frames of 1 spp at 1920x1080 traced as ONE pass each
(overlap 0: one stream, clean per-kernel spans).  T(n) = a + b n per launch: a is what a launch costs before its rays do (ramp, tail), b the steady-state rate.
usage (GPU box): python tools/batch_scaling_probe.py [scene] [option=value ...]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from strelka_amd import capi, scene as S
args = sys.argv[1:]
scene = args[0] if args and "=" not in args[0] else "kitchen"
sc, arr, workload = bench.load_workload(scene)
ctx = capi.Context(0)
ctx.set_option("overlap", 0)
for kv in args:
    if "=" in kv:
        k, v = kv.split("="); ctx.set_option(k, int(v))
ctx.set_scene(arr); W, H = 1920, 1080; ctx.resize(W, H)
depth = 3 if scene.startswith("hair") else 4
image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
cam = sc.getCamera()
ctx.set_option("timing", 1)
rows = []
for n in (1, 2, 4, 8, 16, 32, 64):
    reps = max(2, 32 // n)
    for warm in (True, False):
        ctx.reset_stats()
        for r in range(reps):
            p = np.array(S.frame_params(cam, W, H, subframe_index=0, samples_this_launch=1, spp_total=64, max_depth=depth), copy=True)
            ctx.render_subframes(p, n, image.data_ptr())
        torch.cuda.synchronize()
    st = ctx.stats(); L = st["launches_trace_closest"]
    row = (n, st["ms_trace_closest"] / L * 1e3, st["ms_trace_shadow"] / st["launches_trace_shadow"] * 1e3, st["ms_shade"] / st["launches_shade"] * 1e3, (st["rays_radiance"] + st["rays_shadow"]) / reps / 1e6)
    rows.append(row)
    print("BATCH %s n=%2d: per launch closest %.1f us  any-hit %.1f us  shade %.1f us   (%.2f M rays per call)" % ((scene,) + row))
a = np.array(rows)
for j, name in ((1, "closest"), (2, "any-hit"), (3, "shade")):
    b, a0 = np.polyfit(a[:, 0], a[:, j], 1)
    print("FIT %s %s: %.0f us + %.1f us per sub-frame" % (scene, name, a0, b))
