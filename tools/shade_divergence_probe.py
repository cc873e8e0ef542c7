"""What k_shade's material mix costs: the kitchen with its 60 / 25 / 10 / 5 % diffuse / glossy / metal / glass mix, and with EVERY material replaced by one kind
(the paths differ, the per-path work of k_shade is what is compared)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strelka_amd import capi, scene as S, scenes
import bench
sc, arr, _ = bench.load_workload("kitchen")
W, H, SPP, DEPTH = 1920, 1080, 64, 4
base = arr["materials"].copy()
def run(tag, mats):
    a = dict(arr); a["materials"] = mats
    ctx = capi.Context(0); ctx.set_scene(a); ctx.resize(W, H)
    p = S.frame_params(sc.getCamera(), W, H, subframe_index=0, samples_this_launch=1, spp_total=SPP, max_depth=DEPTH)
    ctx.render_subframes(p, SPP, None)
    ctx.set_option("timing", 1); ctx.reset_stats()
    for _ in range(2): ctx.render_subframes(p, SPP, None)
    st = ctx.stats(); ctx.close()
    rays = st["rays_radiance"] + st["rays_shadow"]
    print("PROBE %-12s shade %.2f ms  closest %.2f  shadow %.2f  rays/frame %.1f M (radiance %.1f M shadow %.1f M)  shade ns/path %.3f" % (tag, st["ms_shade"] / 2, st["ms_trace_closest"] / 2,
          st["ms_trace_shadow"] / 2, rays / 2e6, st["rays_radiance"] / 2e6, st["rays_shadow"] / 2e6, st["ms_shade"] / 2 * 1e6 / (st["rays_radiance"] / 2)), flush=True)
print("types in the mix:", np.bincount(base["type"]), "metallic>0.5:", int((base["metallic"] > 0.5).sum()))
run("mix", base)
for t, name in ((0, "all_diffuse"), (1, "all_pbr"), (2, "all_glass")):
    m = base.copy(); m["type"] = t
    if t == 1: m["roughness"] = np.maximum(m["roughness"], 0.2)
    run(name, m)
