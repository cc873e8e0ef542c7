"""The moving-camera loop (bench.py `interactive`) on its own: ms per call with and without map(), per-kernel spans, for A/B of options.
usage (GPU box): python tools/interactive_probe.py [scene] [option=value ...]   (under rocprofv3 --kernel-trace it gives the per-launch timeline)"""
import sys, time, os, copy, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from strelka_amd import capi, scene as S
args = sys.argv[1:]
scene = args[0] if args and "=" not in args[0] else "kitchen"
opts = [a for a in args if "=" in a]
sc, arr, workload = bench.load_workload(scene)
ctx = capi.Context(0)
calls = 64
for kv in opts:
    k, v = kv.split("=")
    if k == "calls":
        calls = int(v)
    else:
        ctx.set_option(k, int(v))
ctx.set_scene(arr); W, H = 1920, 1080; ctx.resize(W, H)
depth = 3 if scene.startswith("hair") else 4
cam = copy.deepcopy(sc.getCamera())
eye0 = np.array(cam.position, np.float64); fwd = -np.array(cam.rotation[2, :3], np.float64); target = eye0 + fwd * 3.0
image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); host = np.empty((H, W, 4), np.float32); ctx.host_register(host)
def params_at(k):
    a = math.radians(0.25 * k); r = eye0 - target
    e = target + np.array([r[0] * math.cos(a) + r[2] * math.sin(a), r[1], -r[0] * math.sin(a) + r[2] * math.cos(a)])
    cam.lookAt(tuple(e), tuple(target))
    return np.array(S.frame_params(cam, W, H, subframe_index=0, samples_this_launch=1, spp_total=64, max_depth=depth), copy=True)
P = [params_at(k) for k in range(3 * calls + 16)]
for with_map in (True, False):
    for rep in range(2):
        ctx.reset_stats(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(calls):
            ctx.render_subframe(P[rep * calls + k], image.data_ptr())
            if with_map:
                ctx.buffer_download(image.data_ptr(), host)
        dt = time.perf_counter() - t0
    st = ctx.stats(); rays = st["rays_radiance"] + st["rays_shadow"]
    print("LOOP %s map=%d: %.3f ms/call  %.0f Mray/s  (%d rays/call)" % (scene, with_map, dt / calls * 1e3, rays / dt / 1e6, rays / calls))
ctx.set_option("timing", 1); ctx.reset_stats()
for k in range(16):
    ctx.render_subframe(P[2 * calls + k], image.data_ptr())
st = ctx.stats()
print("SPANS per call ms:", {k: round(st[k] / 16, 3) for k in ("ms_trace_closest", "ms_trace_shadow", "ms_shade", "ms_raygen", "ms_accumulate")},
      "launches", {k: st[k] / 16 for k in ("launches_trace_closest", "launches_trace_shadow", "launches_shade", "launches_other")})
