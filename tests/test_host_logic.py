"""Host-side mirror of oka::Scene / Camera / render() parameter logic (strelka_amd/scene.py) against the oracle's
restatement of the same reference code and against hand-derived values."""
import ctypes as C
import math

import numpy as np

from tests.tilehelp import detile_numpy

from strelka_amd import scene as S
from strelka_amd import scenes, tiles


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_pack_normals_and_uv_round_trip_through_the_reference_unpackers(ork):
    rs = np.random.RandomState(0)
    n = rs.normal(size=(500, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    packed = S.pack_normals(n)
    for v, q in zip(n.astype(np.float32), packed):
        v = np.ascontiguousarray(v)
        assert ork.ork_pack_normal(p(v)) == int(q)  # packNormals: scene.cpp:111-117
        out = np.zeros(3, np.float32)
        ork.ork_unpack_normal(int(q), p(out))  # unpackNormal: closest_hit.cu:236-244
        assert np.abs(out - v).max() < 2.0 / 511 + 1e-6
    uv = rs.uniform(-9.9, 9.9, (200, 2)).astype(np.float32)
    for (a, b), q in zip(uv, S.pack_uv(uv)):
        assert ork.ork_pack_uv(float(a), float(b)) == int(q)
        out = np.zeros(2, np.float32)
        ork.ork_unpack_uv(int(q), p(out))
        assert abs(out[0] - a) < 20.0 / 16383 + 1e-5 and abs(out[1] - b) < 20.0 / 16383 + 1e-5
    # the z field is unpacked through a 12-bit mask (0xfff00000) although only 10 bits are packed: harmless because
    # the two extra bits are always zero for packed values
    assert (S.pack_normals([(0, 0, 1)])[0] >> 30) == 0


def test_rect_light_baking_follows_scene_cpp():
    sc = S.Scene()
    xf = S.translate((1.0, 2.0, 3.0)) @ S.rotate((1, 0, 0), math.radians(-90))
    lid = sc.createLight({"type": 0, "xform": xf, "useXform": True, "width": 0.6, "height": 0.4, "color": (1, 2, 3), "intensity": 5.0})
    arr = sc.arrays()
    L = arr["lights"][lid]
    # points = xform * scale(w,h,1) * (+-0.5, +-0.5, 0): scene.cpp:356-369; local +y maps to world -z here
    assert np.allclose(L["points"][0][:3], (1.3, 2.0, 3.0 - 0.2), atol=1e-6)
    assert np.allclose(L["points"][2][:3], (0.7, 2.0, 3.0 + 0.2), atol=1e-6)
    assert np.allclose(L["color"], (5, 10, 15, 5))
    assert L["type"] == 0
    # proxy instance: light type, rect mesh (2 triangles), light id, no material
    inst = arr["instances"][-1]
    assert inst["type"] == S.INSTANCE_LIGHT and inst["light_id"] == lid and inst["material_id"] == 0xFFFFFFFF
    mesh = arr["meshes"][inst["geom_id"]]
    assert mesh["index_count"] == 6 and mesh["vertex_count"] == 4
    # emitting side: calcLightNormal = -normalize(cross(p1-p0, p3-p0)) (Lights.h:54-62) must point down (-y) here
    e1 = L["points"][1][:3] - L["points"][0][:3]
    e2 = L["points"][3][:3] - L["points"][0][:3]
    nrm = -np.cross(e1, e2)
    assert nrm[1] < 0 and abs(nrm[0]) < 1e-6 and abs(nrm[2]) < 1e-6


def test_sphere_disc_distant_light_baking():
    sc = S.Scene()
    sc.createMesh(np.zeros(3, S.VERTEX), [0, 1, 2])  # mesh 0 exists, as in any real scene
    s = sc.createLight({"type": 2, "xform": S.translate((0, 1, 0)), "useXform": True, "radius": 0.25, "color": (1, 1, 1), "intensity": 2.0})
    d = sc.createLight({"type": 3, "xform": np.eye(4), "useXform": True, "halfAngle": 0.1, "color": (1, 1, 1), "intensity": 1.0})
    k = sc.createLight({"type": 1, "xform": np.eye(4), "useXform": True, "radius": 0.5, "color": (1, 1, 1), "intensity": 1.0})
    arr = sc.arrays()
    assert arr["lights"][s]["points"][0][0] == np.float32(0.25) and np.allclose(arr["lights"][s]["points"][1][:3], (0, 1, 0))
    assert np.allclose(arr["lights"][d]["normal"][:3], (0, 0, -1)) and arr["lights"][d]["half_angle"] == np.float32(0.1)
    # the distant light's proxy is an instance of MESH 0 scaled by radius 0: degenerate (scene.cpp:337-345)
    di = arr["instances"][1]
    assert di["type"] == S.INSTANCE_LIGHT and di["geom_id"] == 0 and np.allclose(di["transform"].reshape(3, 4)[:, :3], 0)
    # sphere proxy: 16 x 16 UV sphere = 512 triangles; disc proxy: 16-gon fan = 16 triangles (scene.cpp:153-250)
    assert arr["meshes"][arr["instances"][0]["geom_id"]]["index_count"] == 512 * 3
    assert arr["meshes"][arr["instances"][2]["geom_id"]]["index_count"] == 16 * 3
    assert arr["lights"][k]["type"] == 1


def test_camera_matrices_match_the_reference_formulas(ork):
    cam = S.Camera(fov=39.3, znear=0.1, zfar=1000.0)
    cam.lookAt((1.0, 2.0, 5.0), (0.0, 0.5, 0.0))
    prm = S.frame_params(cam, 640, 480)
    want = np.zeros(16, np.float32)
    ork.ork_clip_to_view(39.3, 640 / 480.0, 0.1, 1000.0, p(want))  # camera.cpp:61-131
    assert np.allclose(prm["clip_to_view"], want, rtol=1e-6, atol=1e-9)
    # centre pixel ray goes from the eye towards the target; view dir is -Z in camera space
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    c2v = np.ascontiguousarray(prm["clip_to_view"])
    v2w = np.ascontiguousarray(prm["view_to_world"])
    ork.ork_camera_ray(320, 240, 640, 480, p(c2v), p(v2w), 0.0, 0.0, p(o), p(d))
    assert np.allclose(o, (1, 2, 5), atol=1e-5)
    t = np.array([0, 0.5, 0]) - np.array([1, 2, 5.0])
    assert np.allclose(d, t / np.linalg.norm(t), atol=1e-5)
    # no y flip: larger pixel y looks further up (image row 0 = bottom; hdRunner flips when writing PNGs)
    d2 = np.zeros(3, np.float32)
    ork.ork_camera_ray(320, 400, 640, 480, p(c2v), p(v2w), 0.0, 0.0, p(o), p(d2))
    assert d2[1] > d[1]


def test_exposure_defaults(ork):
    e = S.default_exposure()
    want = np.zeros(3, np.float32)
    ork.ork_exposure(100.0, 1.0, 4.0, 100.0, p(want))
    assert np.array_equal(e, want) and np.allclose(e, 6.25e-4)
    ork.ork_exposure(0.0, 2.0, 4.0, 100.0, p(want))  # filmIso = 0: "arbitrary" mode, cm2_factor only
    assert np.allclose(S.default_exposure(filmIso=0.0, cm2_factor=2.0), want)


def test_deindex_matches_hdstrelka_mesh_bake():
    pos = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)], np.float32)
    vb, ib = S.deindex(pos, [(0, 1, 2), (0, 2, 3)])
    assert len(vb) == 6 and np.array_equal(ib, np.arange(6))  # 3 fresh vertices per triangle (Mesh.cpp:140-178)
    assert np.array_equal(vb["normal"][:3], S.pack_normals([(0, 0, 1)] * 3))
    assert np.array_equal(vb["normal"][3:], S.pack_normals([(1, 0, 0)] * 3))


def test_scene_generators_meet_the_survey_recipe():
    c = scenes.cornell_box().arrays()
    assert len(c["indices"]) // 3 == 30 + 2 and len(c["instances"]) == 4 and len(c["lights"]) == 1  # 30 tris + light proxy
    k = scenes.kitchen_standin(n_meshes=20, n_instances=120, tri_lo=50, tri_hi=400).arrays()
    assert len(k["lights"]) == 5 and (k["lights"]["type"] == 3).sum() == 1 and (k["lights"]["type"] == 0).sum() == 4
    assert len(k["meshes"]) >= 20 and len(k["instances"]) >= 120
    assert set(np.unique(k["materials"]["type"])) <= {0, 1, 2}
    h = scenes.hair_standin(n_strands=50, n_cp=8).arrays()
    assert len(h["curves"]) == 1 and (h["curve_vertex_counts"] == 10).all()  # +2 phantom points per strand
    assert len(h["curve_points"]) == 500 == len(h["curve_radii"])
    # same seed, same bytes
    a, b = scenes.kitchen_standin(n_meshes=5, n_instances=60, tri_lo=20, tri_hi=60).arrays(), \
        scenes.kitchen_standin(n_meshes=5, n_instances=60, tri_lo=20, tri_hi=60).arrays()
    assert all(np.array_equal(a[key], b[key]) for key in a)


def test_tile_assignment_partitions_the_frame():
    W, H, T = 1920, 1080, 32
    grid = tiles.tile_grid(W, H, T)
    assert len(grid) == 60 * 34
    seen = set()
    for r in range(8):
        mine = tiles.assign_tiles(W, H, T, 8, r)
        assert len(mine) <= tiles.max_tiles_per_rank(W, H, T, 8)
        for t in map(tuple, mine):
            assert t not in seen
            seen.add(t)
    assert len(seen) == len(grid)
    # detile reference: slot order inside a tile is Morton
    data = np.arange(2 * 16 * 16 * 4, dtype=np.float32).reshape(2, 256, 4)
    img = detile_numpy(data, np.array([(0, 0), (16, 0)]), 16, 24, 16)
    assert img[0, 0, 0] == 0 and img[0, 1, 0] == 4 and img[1, 0, 0] == 8 and img[1, 1, 0] == 12  # z-order
    assert img[0, 16, 0] == 256 * 4 and (img[:, 24:] == 0).all() if img.shape[1] > 24 else True
