"""Would two half-frame pipelines side by side shorten the moving-camera call?  Two CONTEXTS (own streams, queues, hierarchy copies), each owning half of the
32 x 32 tiles (alternating), driven by two host threads, one 1-spp 1080p skh_render_subframe each per frame, against one context rendering the whole frame.
usage (GPU box): python tools/split_probe.py [scene] [parts=2] [option=value ...]"""
import sys, time, os, copy, math, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from strelka_amd import capi, scene as S
args = sys.argv[1:]
scene = args[0] if args and "=" not in args[0] else "kitchen"
opts = dict(a.split("=") for a in args if "=" in a)
parts = int(opts.pop("parts", 2)); calls = int(opts.pop("calls", 64)); tile = int(opts.pop("tile", 32))
sc, arr, workload = bench.load_workload(scene)
W, H = 1920, 1080
depth = 3 if scene.startswith("hair") else 4
cam = copy.deepcopy(sc.getCamera())
eye0 = np.array(cam.position, np.float64); fwd = -np.array(cam.rotation[2, :3], np.float64); target = eye0 + fwd * 3.0
def params_at(k):
    a = math.radians(0.25 * k); r = eye0 - target
    e = target + np.array([r[0] * math.cos(a) + r[2] * math.sin(a), r[1], -r[0] * math.sin(a) + r[2] * math.cos(a)])
    cam.lookAt(tuple(e), tuple(target))
    return np.array(S.frame_params(cam, W, H, subframe_index=0, samples_this_launch=1, spp_total=64, max_depth=depth), copy=True)
P = [params_at(k) for k in range(2 * calls)]
image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
def make(n, i):
    ctx = capi.Context(0)
    for k, v in opts.items():
        ctx.set_option(k, int(v))
    ctx.set_scene(arr)
    if n > 1:
        tx, ty = (W + tile - 1) // tile, (H + tile - 1) // tile
        own = [(x * tile, y * tile) for y in range(ty) for x in range(tx) if (y * tx + x) % n == i]
        ctx.set_tiles(tile, np.array(own, np.uint32))
    ctx.resize(W, H)
    return ctx
def loop(ctxs):
    n = len(ctxs); bar = threading.Barrier(n + 1) if n > 1 else None; out = {}
    def worker(i):
        for rep in range(2):
            for k in range(calls):
                bar.wait(); ctxs[i].render_subframe(P[rep * calls + k], image.data_ptr()); bar.wait()
    if n > 1:
        th = [threading.Thread(target=worker, args=(i,)) for i in range(n)]; [t.start() for t in th]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(calls):
            if n > 1:
                bar.wait(); bar.wait()
            else:
                ctxs[0].render_subframe(P[rep * calls + k], image.data_ptr())
        out[rep] = (time.perf_counter() - t0) / calls * 1e3
    if n > 1:
        [t.join() for t in th]
    return out[1]
one = make(1, 0); t1 = loop([one]); ref = image.clone(); del one
cs = [make(parts, i) for i in range(parts)]; tn = loop(cs)
same = bool((image == ref).all())
print("SPLIT %s: one context %.3f ms/call; %d contexts side by side %.3f ms/call; images equal: %s" % (scene, t1, parts, tn, same))
