"""Flat binary scene dump: strelka_amd/scene_io.py <-> oka::Scene::saveDump / loadDump (strelka_amd/host).

The dump carries exactly the arrays the renderer uploads, so the checks are byte equalities: Python round trip, the C++
writer read by the Python reader (against the Python restatement of the same scene recipe), the Python writer read and
re-written by the C++ reader, malformed files rejected, MaterialDescription -> skh_material mapping."""
import os
import struct

import numpy as np
import pytest

from strelka_amd import scene as S, scene_io, scenes
from tests.test_host_cpp import python_recipe, run_host


def flat_bytes(v):
    """array, or list of arrays (textures): shape-tagged bytes for equality checks"""
    return v.tobytes() if hasattr(v, "tobytes") else b"".join(repr(t.shape).encode() + t.tobytes() for t in v)


def same(a, b):
    assert set(a) == set(b)
    for k in a:
        if isinstance(a[k], list):
            assert flat_bytes(a[k]) == flat_bytes(b[k]), k
        else:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), k


def test_python_round_trip_with_curves(tmp_path):
    sc = scenes.hair_standin(n_strands=300)
    arr = sc.arrays()
    path = os.path.join(tmp_path, "hair.skscene")
    scene_io.save_scene(path, arr, sc.getCamera())
    back = scene_io.load_scene(path)
    same(arr, back.arrays())
    cam = back.getCamera()
    assert np.allclose(cam.view, sc.getCamera().view, atol=1e-6) and cam.fov == pytest.approx(sc.getCamera().fov)
    # the loaded camera produces the same frame parameters
    a = S.frame_params(sc.getCamera(), 64, 48)
    b = S.frame_params(cam, 64, 48)
    assert np.allclose(a["view_to_world"], b["view_to_world"], atol=1e-5) and np.array_equal(a["clip_to_view"], b["clip_to_view"])


def test_cpp_writer_is_read_by_python_and_equals_the_python_recipe(tmp_path):
    run_host(tmp_path, "cpu")
    got = scene_io.load_scene(os.path.join(tmp_path, "scene.skscene"))
    want = python_recipe()
    wa, ga = want.arrays(), got.arrays()
    for k in ("vertices", "indices", "meshes", "instances", "materials"):
        if k == "instances":
            assert np.allclose(wa[k]["transform"], ga[k]["transform"], atol=1e-6)
            for f in ("type", "geom_id", "material_id", "light_id"):
                assert np.array_equal(wa[k][f], ga[k][f])
        else:
            assert wa[k].tobytes() == ga[k].tobytes(), k
    assert np.allclose(wa["lights"]["points"], ga["lights"]["points"], atol=1e-5)
    assert np.allclose(got.getCamera().view, want.getCamera().view, atol=1e-5)


def test_python_writer_is_read_and_rewritten_identically_by_cpp(tmp_path):
    sc = scenes.cornell_box()
    src = os.path.join(tmp_path, "cornell.skscene")
    scene_io.save_scene(src, sc.arrays(), sc.getCamera())
    out = run_host(tmp_path, "load", src)
    assert "load ok" in out
    a, b = scene_io.load_scene(src), scene_io.load_scene(os.path.join(tmp_path, "resaved.skscene"))
    same(a.arrays(), b.arrays())
    assert np.allclose(a.getCamera().view, b.getCamera().view, atol=1e-5)  # (through position + quaternion and back)


def test_malformed_dumps_are_rejected(tmp_path):
    sc = scenes.cornell_box()
    good = os.path.join(tmp_path, "ok.skscene")
    scene_io.save_scene(good, sc.arrays(), sc.getCamera())
    blob = open(good, "rb").read()
    cases = {"magic": b"NOTSCENE" + blob[8:], "truncated": blob[:len(blob) // 2], "version": blob[:8] + struct.pack("<I", 9) + blob[12:]}
    arr = {k: v.copy() for k, v in sc.arrays().items()}
    arr["meshes"]["index_count"][0] = 10 ** 6  # mesh reaching outside the index buffer
    bad = os.path.join(tmp_path, "range.skscene")
    scene_io.save_scene(bad, arr, sc.getCamera())
    cases["range"] = open(bad, "rb").read()
    from strelka_amd import build
    import subprocess

    exe = build.build_host()
    for name, data in cases.items():
        p = os.path.join(tmp_path, name + ".skscene")
        open(p, "wb").write(data)
        with pytest.raises(ValueError):
            scene_io.load_scene(p)
        r = subprocess.run([exe, "load", str(tmp_path), p], capture_output=True, text=True)
        assert r.returncode != 0, name


def test_material_descriptions_map_to_the_argument_block(tmp_path):
    f3 = lambda v: [float(x) for x in v]
    descs = [
        {"file": "default.mdl", "name": "default_material", "params": [{"name": "diffuse_color", "type": "float3", "value": f3((0.1, 0.2, 0.3))}]},
        {"file": "OmniPBR.mdl", "name": "OmniPBR", "params": [  # gltfloader.cpp:304-352
            {"name": "diffuse_color_constant", "type": "float3", "value": f3((0.9, 0.5, 0.1))},
            {"name": "reflection_roughness_constant", "type": "float", "value": 0.25},
            {"name": "metallic_constant", "type": "float", "value": 1.0}]},
        {"file": "OmniGlass.mdl", "name": "OmniGlass", "params": [  # gltfloader.cpp:354-406
            {"name": "enable_opacity", "type": "bool", "value": True}, {"name": "thin_walled", "type": "bool", "value": False},
            {"name": "frosting_roughness", "type": "float", "value": 0.0}]},
        {"file": "unknown.mdl", "name": "something_else", "params": []},
        {"file": "", "name": "Kitchen_set_Material_12", "params": [  # UsdPreviewSurface via HdStrelka (Material.cpp:52-150)
            {"name": "diffuseColor", "type": "float3", "value": f3((0.3, 0.25, 0.2))}, {"name": "roughness", "type": "float", "value": 0.7},
            {"name": "metallic", "type": "float", "value": 0.0}, {"name": "useSpecularWorkflow", "type": "int", "value": 0}]},
        {"file": "", "name": "window_glass", "params": [{"name": "diffuseColor", "type": "float3", "value": f3((1, 1, 1))},
                                                         {"name": "opacity", "type": "float", "value": 0.1}, {"name": "ior", "type": "float", "value": 1.45}]},
    ]
    m = scene_io.materials_from_descriptions(descs)
    assert list(m["type"]) == [S.MAT_DIFFUSE, S.MAT_PBR, S.MAT_GLASS, S.MAT_DIFFUSE, S.MAT_PBR, S.MAT_GLASS]
    assert np.allclose(m["base_color"][4], (0.3, 0.25, 0.2)) and m["roughness"][4] == np.float32(0.7) and m["ior"][5] == np.float32(1.45)
    assert np.allclose(m["base_color"][0], (0.1, 0.2, 0.3)) and np.allclose(m["base_color"][1], (0.9, 0.5, 0.1))
    assert m["roughness"][1] == np.float32(0.25) and m["metallic"][1] == 1.0 and m["ior"][2] == np.float32(1.491)
    assert np.allclose(m["base_color"][3], 0.8)
    # a dump that carries descriptions instead of argument blocks
    sc = scenes.cornell_box()
    arr = dict(sc.arrays())
    arr.pop("materials")
    p = os.path.join(tmp_path, "descs.skscene")
    scene_io.save_scene(p, arr, sc.getCamera(), material_descriptions=descs)
    back = scene_io.load_scene(p)
    assert back.arrays()["materials"].tobytes() == m.tobytes() and back.material_descriptions == descs
