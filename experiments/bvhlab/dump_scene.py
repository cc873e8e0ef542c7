"""bvhlab input: the world-space triangles of a bench scene (every mesh instance baked, as bake_world 4 does) and a set of camera rays,
as raw float32 files.  Usage: python experiments/bvhlab/dump_scene.py kitchen_arch /tmp/lab_arch"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from strelka_amd import scene as S, scenes  # noqa: E402
from tests.test_gpu_parity import camera_rays  # noqa: E402


def main():
    name, out = sys.argv[1], sys.argv[2]
    if name == "kitchen_arch":
        sc = scenes.kitchen_architectural()
    elif name == "kitchen":
        sc = scenes.kitchen_standin()
    elif name == "kitchen_small":
        sc = scenes.kitchen_standin(seed=1234, n_meshes=40, n_instances=300, tri_lo=200, tri_hi=8000)
    else:
        raise SystemExit(name)
    arr = sc.arrays()
    pos = arr["vertices"]["pos"]
    chunks = []
    for inst in arr["instances"]:
        if inst["type"] != S.INSTANCE_MESH:
            continue
        m = arr["meshes"][inst["geom_id"]]
        idx = arr["indices"][m["index_offset"]:m["index_offset"] + m["index_count"]].astype(np.int64) + int(m["vertex_offset"])
        p = pos[idx].astype(np.float32)
        t = inst["transform"].reshape(3, 4).astype(np.float32)
        w = (p @ t[:, :3].T + t[:, 3]).astype(np.float32)
        chunks.append(w.reshape(-1, 9))
    tris = np.concatenate(chunks)
    tris.tofile(out + ".tris")
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
    r = camera_rays(sc, 1920, 1080, n, 3)
    np.concatenate([r["origin"], r["dir"]], 1).astype(np.float32).tofile(out + ".rays")
    print(len(tris), "triangles,", n, "camera rays")


main()
