"""One render() per sub-frame (the reference caller's pattern): per-kernel spans of a few calls + the loop rate.
usage (GPU box): python tools/interactive_loop.py [option=value ...]   e.g. speculate=0, overlap=0"""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
from strelka_amd import capi, scene as S, scenes
sc = scenes.kitchen_standin(); arr = sc.arrays()
ctx = capi.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
ctx.set_scene(arr); W, H = 1920, 1080; ctx.resize(W, H)
img = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
p = np.array(S.frame_params(sc.getCamera(), W, H, subframe_index=0, samples_this_launch=1, spp_total=64, max_depth=4), copy=True)
for i in range(8):
    p["subframe_index"] = i; ctx.render_subframe(p, img.data_ptr())
ctx.set_option("timing", 1); ctx.reset_stats()
for i in range(8, 24):
    p["subframe_index"] = i; ctx.render_subframe(p, img.data_ptr())
st = ctx.stats()
print("SPANS per subframe ms:", {k: round(st[k] / 16, 3) for k in ("ms_trace_closest", "ms_trace_shadow", "ms_shade", "ms_raygen", "ms_accumulate")}, "sum", round(sum(st[k] for k in ("ms_trace_closest", "ms_trace_shadow", "ms_shade", "ms_raygen", "ms_accumulate")) / 16, 3))
ctx.set_option("timing", 0); ctx.reset_stats()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(64):
    p["subframe_index"] = i; ctx.render_subframe(p, img.data_ptr())
dt = time.perf_counter() - t0
st = ctx.stats(); rays = st["rays_radiance"] + st["rays_shadow"]
print("LOOP", sys.argv[1:], "ms/subframe %.3f  Mray/s %.0f" % (dt / 64 * 1e3, rays / dt / 1e6))
