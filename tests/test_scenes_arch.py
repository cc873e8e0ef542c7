"""kitchen_architectural(): the less forgiving C3 stand-in must really have what it is there for -- no mesh sharing,
triangles metres long next to centimetre-sized ones, long thin triangles, nested containment -- and stay seeded."""
import numpy as np

from strelka_amd import scene as S, scenes


def _world_triangles(arr):
    v, idx = arr["vertices"]["pos"], arr["indices"]
    out = []
    for inst in arr["instances"]:
        if inst["type"] != S.INSTANCE_MESH:
            continue
        m = arr["meshes"][inst["geom_id"]]
        tri = idx[m["index_offset"]:m["index_offset"] + m["index_count"]].reshape(-1, 3) + m["vertex_offset"]
        p = v[tri].astype(np.float64)  # (n, 3, 3)
        M = np.asarray(inst["transform"], np.float64).reshape(3, 4)
        out.append(p @ M[:, :3].T + M[:, 3])
    return np.concatenate(out)


def test_architectural_kitchen_has_the_geometry_it_promises():
    sc = scenes.kitchen_architectural()
    arr = sc.arrays()
    ntri = len(arr["indices"]) // 3
    mesh_inst = arr["instances"][arr["instances"]["type"] == S.INSTANCE_MESH]
    assert 1.5e6 < ntri < 1.8e6 and len(mesh_inst) > 2000
    # one mesh per instance: what HdStrelka's bake hands over (RenderPass.cpp:126-129,252-257)
    assert len(np.unique(mesh_inst["geom_id"])) == len(mesh_inst)
    t = _world_triangles(arr)
    e = np.stack([np.linalg.norm(t[:, 1] - t[:, 0], axis=1), np.linalg.norm(t[:, 2] - t[:, 1], axis=1), np.linalg.norm(t[:, 0] - t[:, 2], axis=1)], 1)
    area = 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1)
    longest = e.max(1)
    aspect = longest * longest / np.maximum(2.0 * area, 1e-30)  # longest edge / height over it
    assert (longest > 3.0).sum() >= 40  # walls, floor, counter tops, pipes: triangles metres long ...
    assert np.median(longest) < 0.02  # ... among crockery triangles of a centimetre
    assert (aspect > 30).sum() >= 3000  # rods, slats, panel edges: long thin triangles
    # nested containment: crockery boxes lie inside cabinet / counter compartments, which lie inside the room
    lo, hi = t.min(axis=(0, 1)), t.max(axis=(0, 1))
    assert np.all(lo >= [-5.001, -0.001, -3.001]) and np.all(hi <= [5.001, 4.001, 3.001])
    # seeded: the same scene every time
    arr2 = scenes.kitchen_architectural().arrays()
    assert np.array_equal(arr["vertices"]["pos"], arr2["vertices"]["pos"]) and np.array_equal(arr["instances"]["transform"], arr2["instances"]["transform"])
