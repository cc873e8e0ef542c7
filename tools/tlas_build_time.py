"""Build time of the two TLAS builders over the instance count.  usage (GPU box): python tools/tlas_build_time.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from strelka_amd import capi, scene as S, scenes
for N in (2000, 20000, 100000):
    rs = np.random.RandomState(9)
    sc = S.Scene(); mat = sc.addMaterial(S.MAT_DIFFUSE, (0.7, 0.7, 0.7))
    pos, tris = scenes._grid_mesh(scenes._sphere_fn(rs, 0.1), 5, 4); mesh = scenes._add_mesh(sc, pos, tris)
    P = rs.uniform(-40, 40, (N, 3))
    for k in range(N):
        sx = rs.uniform(0.1, 0.4); sc.createInstance(S.INSTANCE_MESH, mesh, mat, S.translate(P[k]) @ S.scale((sx, sx, sx)))
    arr = sc.arrays()
    for mode in (1, 0):
        if mode == 0 and N > 20000:
            continue
        ctx = capi.Context(0); ctx.set_option("tlas_build", mode); ctx.set_scene(arr); ctx.set_scene(arr)
        print("TLAS", N, "instances, builder", "gpu-ploc" if mode else "host-sah", "skh_build_accel %.1f ms" % ctx.stats()["ms_build"], flush=True)
        ctx.close()
