"""What would octant-binned compaction buy the traversal?  (round-2 review, item 1: "octant-binned compaction ... measure alone")
Secondary rays as the pipeline sees them at bounce 1 -- origins = the primary hit points of a 1080p frame in the queue's order
(tile-major, Morton inside a tile: neighbouring pixels adjacent), directions = cosine-weighted about a random axis -- are traced by
skh_trace_device in three orders: (a) as they are, (b) stably partitioned by direction octant inside runs of 256 K rays (what 8 tails
per shard would produce), (c) octant + 10-bit Morton code of the origin inside the same runs (an upper bound for any cheap binning).
Same rays, same hits; only the order differs.  usage (GPU box): python tools/octant_bin_probe.py [kitchen|kitchen_unshared] [spp]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from strelka_amd import capi, scene as S, tiles

name = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc, arr, line = bench.load_workload(name)
W, H, T = 1920, 1080, 32
ctx = capi.Context(0)
ctx.set_scene(arr)
# primary rays in slot order (tile-major, Morton inside the tile), `spp` jittered samples per pixel
rs = np.random.RandomState(1)
cam = sc.getCamera()
p = S.frame_params(cam, W, H)
v2w = p["view_to_world"].reshape(4, 4).astype(np.float64)
c2v = p["clip_to_view"].reshape(4, 4).astype(np.float64)
def morton_order(t):
    y, x = np.mgrid[0:t, 0:t]
    def part(v):
        v = v.astype(np.uint32); v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555; return v
    code = part(x) | (part(y) << 1)
    o = np.argsort(code.reshape(-1))
    return x.reshape(-1)[o], y.reshape(-1)[o]
mx, my = morton_order(T)
px, py = [], []
for ty in range(0, H, T):
    for tx in range(0, W, T):
        xs, ys = tx + mx, ty + my
        ok = (xs < W) & (ys < H)
        px.append(xs[ok]); py.append(ys[ok])
px, py = np.concatenate(px), np.concatenate(py)
px, py = np.tile(px, spp), np.tile(py, spp)
n = len(px)
jx, jy = rs.rand(n), rs.rand(n)
ndc = np.stack([(px + jx) / W * 2 - 1, (py + jy) / H * 2 - 1, np.ones(n), np.ones(n)], 1)
view = ndc @ c2v.T
d = np.concatenate([view[:, :3], np.zeros((n, 1))], 1) @ v2w.T
d = d[:, :3] / np.linalg.norm(d[:, :3], axis=1, keepdims=True)
prim = np.zeros(n, S.RAY); prim["origin"] = v2w[:3, 3]; prim["dir"] = d; prim["tmax"] = 1e16
hits = ctx.trace(prim, 0)
ok = hits["instance_id"] != 0xFFFFFFFF
P = prim["origin"][ok].astype(np.float64) + prim["dir"][ok].astype(np.float64) * hits["t"][ok][:, None].astype(np.float64)
m = len(P)
dirs = rs.normal(size=(m, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
sec = np.zeros(m, S.RAY)
sec["origin"] = P - prim["dir"][ok] * 1e-3  # step back off the surface
sec["dir"] = dirs; sec["tmax"] = 1e16
octant = (dirs[:, 0] < 0) * 1 + (dirs[:, 1] < 0) * 2 + (dirs[:, 2] < 0) * 4
RUN = 1 << 18
def binned(key):
    order = np.arange(m)
    for a in range(0, m, RUN):
        b = min(m, a + RUN)
        order[a:b] = a + np.argsort(key[a:b], kind="stable")
    return order
lo, hi = P.min(0), P.max(0)
q = np.clip(((P - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint32), 0, 1023)
def part3(v):
    v = v.astype(np.uint64); v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
mort = part3(q[:, 0]) | (part3(q[:, 1]) << 1) | (part3(q[:, 2]) << 2)
orders = {"queue order": np.arange(m), "octant bins per 256 K run": binned(octant), "octant + origin Morton per run": binned(octant.astype(np.uint64) << 30 | mort)}
ref = None
for label, order in orders.items():
    rays = np.ascontiguousarray(sec[order])
    d_r = torch.from_numpy(rays.view(np.uint8).copy()).cuda()
    d_h = torch.zeros(m * 20, dtype=torch.uint8, device="cuda")
    for mode, kind in ((0, "closest"), (1, "any-hit")):
        if mode == 1:
            r2 = rays.copy(); r2["tmax"] = 3.0
            d_r = torch.from_numpy(r2.view(np.uint8).copy()).cuda()
        ctx.trace_device(d_r.data_ptr(), m, mode, d_h.data_ptr(), 1)  # warm-up
        ctx.set_option("timing", 1); ctx.reset_stats()
        ctx.trace_device(d_r.data_ptr(), m, mode, d_h.data_ptr(), 4)
        st = ctx.stats(); ctx.set_option("timing", 0)
        ms = (st["ms_trace_shadow"] if mode else st["ms_trace_closest"]) / 4
        print("OCTANT %-18s %-32s %-8s %8d rays  %.3f ms  %.0f Mray/s" % (name, label, kind, m, ms, m / ms / 1e3), flush=True)
ctx.close()

# ---- shadow rays as the renderer makes them: from the primary hit points to a random point on one of the scene's rect lights (or along the
#      distant light), in queue order vs re-ordered by light INSIDE every 256-ray block (what a block-local binned compaction in k_shade would write)
ctx = capi.Context(0)
ctx.set_scene(arr)
L = arr["lights"]
nl = len(L)
lid = rs.randint(0, nl, m)
sh = np.zeros(m, S.RAY)
sh["origin"] = sec["origin"]
tgt = np.zeros((m, 3))
far = np.zeros(m, bool)
for k in range(nl):
    sel = lid == k
    pts = L["points"][k][:, :3].astype(np.float64)  # 4 corners (rect) -- good enough for every type here
    if int(L["type"][k]) == 0:
        u, v = rs.rand(sel.sum(), 1), rs.rand(sel.sum(), 1)
        tgt[sel] = pts[0] + (pts[1] - pts[0]) * u + (pts[3] - pts[0]) * v
    else:
        tgt[sel] = sh["origin"][sel] + np.array([0.35, 0.8, 0.45]) * 1e3
        far[sel] = True
dv = tgt - sh["origin"]
ln = np.linalg.norm(dv, axis=1, keepdims=True)
sh["dir"] = dv / ln
sh["tmax"] = np.where(far, 1e9, ln[:, 0] * 0.999)
blk = np.arange(m) // 256
orders = {"queue order": np.arange(m), "by light inside 256-ray blocks": np.lexsort((np.arange(m), lid, blk)), "by light inside 1024-ray blocks": np.lexsort((np.arange(m), lid, np.arange(m) // 1024))}
for label, order in orders.items():
    rays = np.ascontiguousarray(sh[order])
    d_r = torch.from_numpy(rays.view(np.uint8).copy()).cuda()
    d_h = torch.zeros(m * 20, dtype=torch.uint8, device="cuda")
    ctx.trace_device(d_r.data_ptr(), m, 1, d_h.data_ptr(), 1)
    ctx.set_option("timing", 1); ctx.reset_stats()
    ctx.trace_device(d_r.data_ptr(), m, 1, d_h.data_ptr(), 4)
    ms = ctx.stats()["ms_trace_shadow"] / 4; ctx.set_option("timing", 0)
    occ = (torch.frombuffer(bytearray(d_h.cpu().numpy().tobytes()), dtype=torch.float32).reshape(-1, 5)[:, 0] > 0).float().mean().item()
    print("OCTANT %-18s %-32s %-8s %8d rays  %.3f ms  %.0f Mray/s  occluded %.2f" % (name, label, "NEE", m, ms, m / ms / 1e3, occ), flush=True)
ctx.close()
