"""glTF loader (strelka_amd/gltf.py) against the semantics of the reference's src/sceneloader/gltfloader.cpp, on a small
file generated here: node transforms (TRS and matrix, parent * local), one mesh + instance per primitive, index widths,
material conversion (OPAQUE -> OmniPBR, else OmniGlass), material -1 -> 0, the last-triangle tangent quirk, camera
conversion, the default distant light and `<model>_light.json`."""
import base64
import json
import math
import os

import numpy as np
import pytest

from strelka_amd import gltf, scene as S, scene_io


def flat_bytes(v):
    """array, or list of arrays (textures): shape-tagged bytes for equality checks"""
    return v.tobytes() if hasattr(v, "tobytes") else b"".join(repr(t.shape).encode() + t.tobytes() for t in v)


def _pad4(b):
    return b + b"\0" * (-len(b) % 4)


def make_gltf(tmp_path, embed=True, with_lights=False, name="model"):
    quad_p = np.array([(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)], np.float32)
    quad_n = np.array([(0, 2, 0)] * 4, np.float32)  # not unit length: the loader normalises (gltfloader.cpp:147)
    quad_uv = np.array([(0, 0), (1, 0), (1, 1), (0, 1)], np.float32)
    quad_i = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    tri_p = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0)], np.float32)
    tri_i8 = np.array([0, 1, 2, 2, 1, 3], np.uint8)
    tri_i32 = np.array([0, 1, 2], np.uint32)
    chunks, views = [], []

    def add(arr):
        off = sum(len(c) for c in chunks)
        raw = _pad4(arr.tobytes())
        chunks.append(raw)
        views.append({"buffer": 0, "byteOffset": off, "byteLength": arr.nbytes})
        return len(views) - 1

    v = [add(a) for a in (quad_p, quad_n, quad_uv, quad_i, tri_p, tri_i8, tri_i32)]
    acc = [
        {"bufferView": v[0], "componentType": 5126, "count": 4, "type": "VEC3"},
        {"bufferView": v[1], "componentType": 5126, "count": 4, "type": "VEC3"},
        {"bufferView": v[2], "componentType": 5126, "count": 4, "type": "VEC2"},
        {"bufferView": v[3], "componentType": 5123, "count": 6, "type": "SCALAR"},
        {"bufferView": v[4], "componentType": 5126, "count": 4, "type": "VEC3"},
        {"bufferView": v[5], "componentType": 5121, "count": 6, "type": "SCALAR"},
        {"bufferView": v[6], "componentType": 5125, "count": 3, "type": "SCALAR"},
    ]
    blob = b"".join(chunks)
    q = S.quat_from_euler_deg((0.0, 30.0, 0.0))  # w x y z
    mtx = (S.translate((0.5, 0.25, -2.0)) @ S.scale((2.0, 2.0, 2.0)))
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 2]}],
        "nodes": [
            {"name": "root", "translation": [1.0, 2.0, 3.0], "rotation": [q[1], q[2], q[3], q[0]], "scale": [2.0, 1.0, 0.5], "mesh": 0,
             "children": [1]},
            {"name": "child", "matrix": [float(x) for x in mtx.T.reshape(16)], "mesh": 1},
            {"name": "cam", "translation": [0.0, 2.5, 6.0], "rotation": [0.0, math.sin(-0.2), 0.0, math.cos(-0.2)], "camera": 0},
        ],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]},
                   {"primitives": [{"attributes": {"POSITION": 4}, "indices": 5}, {"attributes": {"POSITION": 4}, "indices": 6, "material": 1}]}],
        "materials": [
            {"name": "red", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.1, 0.05, 1.0], "roughnessFactor": 0.4, "metallicFactor": 0.0,
                                                     "baseColorTexture": {"index": 0}}},
            {"name": "pane", "alphaMode": "BLEND", "pbrMetallicRoughness": {"roughnessFactor": 0.1}},
        ],
        "textures": [{"source": 0}], "images": [{"uri": "albedo.png"}],
        "cameras": [{"type": "perspective", "name": "main", "perspective": {"yfov": 0.6, "znear": 0.05, "zfar": 500.0}}],
        "accessors": acc, "bufferViews": views,
        "buffers": [{"byteLength": len(blob), "uri": ("data:application/octet-stream;base64," + base64.b64encode(blob).decode()) if embed else name + ".bin"}],
    }
    path = os.path.join(tmp_path, name + ".gltf")
    with open(path, "w") as f:
        json.dump(doc, f)
    if not embed:
        open(os.path.join(tmp_path, name + ".bin"), "wb").write(blob)
    if with_lights:
        json.dump({"lights": [{"position": [0, 3, 0], "orientation": [-90, 0, 0], "width": 1.5, "height": 0.5, "color": [1, 0.9, 0.8],
                               "intensity": 50}]}, open(os.path.join(tmp_path, name + "_light.json"), "w"))
    return path, dict(quad_p=quad_p, quad_uv=quad_uv, tri_p=tri_p, q=q, mtx=mtx)


def test_nodes_meshes_instances_and_index_widths(tmp_path):
    path, ref = make_gltf(tmp_path)
    sc = gltf.load_gltf(path)
    arr = sc.arrays()
    # light first (default distant light: instance of mesh 0 scaled by radius 0), then one mesh + instance per primitive
    assert len(arr["meshes"]) == 3 and len(arr["instances"]) == 4
    assert list(arr["meshes"]["index_count"]) == [6, 6, 3] and list(arr["meshes"]["vertex_count"]) == [4, 4, 4]
    assert list(arr["instances"]["type"]) == [S.INSTANCE_LIGHT, S.INSTANCE_MESH, S.INSTANCE_MESH, S.INSTANCE_MESH]
    assert list(arr["instances"]["material_id"][1:]) == [0, 0, 1]  # primitive without material -> 0 (gltfloader.cpp:134-138)
    root = S.translate((1, 2, 3)) @ S.quat_to_mat4(ref["q"]) @ S.scale((2.0, 1.0, 0.5))
    child = root @ ref["mtx"]
    assert np.allclose(arr["instances"]["transform"][1].reshape(3, 4), root[:3], atol=1e-6)
    assert np.allclose(arr["instances"]["transform"][2].reshape(3, 4), child[:3], atol=1e-6)
    assert np.allclose(arr["instances"]["transform"][3].reshape(3, 4), child[:3], atol=1e-6)
    assert np.array_equal(arr["indices"][:6], [0, 1, 2, 0, 2, 3]) and np.array_equal(arr["indices"][6:12], [0, 1, 2, 2, 1, 3])
    assert np.array_equal(arr["indices"][12:], [0, 1, 2])
    v = arr["vertices"]
    assert np.array_equal(v["pos"][:4], ref["quad_p"]) and np.array_equal(v["pos"][4:8], ref["tri_p"])
    assert np.array_equal(v["normal"][:4], S.pack_normals([(0, 1, 0)] * 4))  # normalised before packing
    assert np.array_equal(v["uv"][:4], S.pack_uv(ref["quad_uv"]))
    assert np.array_equal(v["normal"][4:8], S.pack_normals([(0, 0, 1)] * 4))  # no NORMAL: geometric normal (stated deviation)
    assert np.array_equal(v["uv"][4:8], S.pack_uv([(0, 0)] * 4))


def test_tangent_is_computed_for_the_last_triangle_only(tmp_path):
    path, ref = make_gltf(tmp_path)
    v = gltf.load_gltf(path).arrays()["vertices"]
    # quad: last triangle = (0, 2, 3): those three vertices share one packed tangent, vertex 1 keeps 0 (gltfloader.cpp:64-93)
    assert v["tangent"][1] == 0 and v["tangent"][0] == v["tangent"][2] == v["tangent"][3] != 0
    # expected value from the loader's own formula: its unpackUV (/16383.99999 * 10 - 5) of the packed uvs
    uv = [gltf._unpack_uv_loader(S.pack_uv([ref["quad_uv"][i]])[0]) for i in (0, 2, 3)]
    p = [ref["quad_p"][i] for i in (0, 2, 3)]
    e1, e2 = uv[1] - uv[0], uv[2] - uv[0]
    d = e1[0] * e2[1] - e1[1] * e2[0]
    t = ((p[1] - p[0]) * e2[1] - (p[2] - p[0]) * e1[1]) / d
    assert v["tangent"][0] == gltf.pack_tangent(t.astype(np.float32))[0]


def test_materials_cameras_and_default_light(tmp_path):
    path, _ = make_gltf(tmp_path)
    sc = gltf.load_gltf(path)
    d = sc.material_descriptions
    assert [m["name"] for m in d] == ["OmniPBR", "OmniGlass"]
    p = {x["name"]: x["value"] for x in d[0]["params"]}
    assert p["diffuse_color_constant"] == [0.8, 0.1, 0.05] and p["reflection_roughness_constant"] == 0.4 and p["metallic_constant"] == 0.0
    assert p["diffuse_texture"] == "albedo.png"
    g = {x["name"]: x["value"] for x in d[1]["params"]}
    assert g == {"enable_opacity": True, "thin_walled": False, "frosting_roughness": 0.1}
    m = sc.arrays()["materials"]
    assert list(m["type"]) == [S.MAT_PBR, S.MAT_GLASS] and np.allclose(m["base_color"][0], (0.8, 0.1, 0.05))
    # camera: fov = yfov * (180 / 3.1415926); node transform -> position, conjugated rotation (gltfloader.cpp:276-291, 422-451)
    cam = sc.getCamera(0)
    assert cam.fov == pytest.approx(0.6 * 180.0 / 3.1415926, rel=1e-6) and cam.znear == pytest.approx(0.05) and cam.zfar == 500.0
    assert np.allclose(cam.position, (0, 2.5, 6))
    rot = S.quat_to_mat4((math.cos(-0.2), 0.0, math.sin(-0.2), 0.0))
    assert np.allclose(cam.view, rot.T @ S.translate((0, -2.5, -6)), atol=1e-6)
    # default distant light (gltfloader.cpp:664-678)
    L = sc.arrays()["lights"]
    assert len(L) == 1 and L["type"][0] == 3 and L["half_angle"][0] == np.float32(10.0 * 0.5 * (math.pi / 180.0))
    assert np.allclose(L["color"][0], (100000, 100000, 100000, 100000))
    n = S.quat_to_mat4(S.quat_from_euler_deg((-45.0, 15.0, 0.0))) @ np.array([0, 0, -1.0, 0])
    assert np.allclose(L["normal"][0], n, atol=1e-6)


def test_light_file_and_external_buffer(tmp_path):
    path, _ = make_gltf(tmp_path, embed=False, with_lights=True, name="ext")
    sc = gltf.load_gltf(path)
    arr = sc.arrays()
    L = arr["lights"]
    assert len(L) == 1 and L["type"][0] == 0 and np.allclose(L["color"][0][:3], np.array([1, 0.9, 0.8]) * 50)
    want = S.Scene()
    want.createLight({"type": 0, "useXform": False, "position": (0, 3, 0), "orientation": (-90, 0, 0), "width": 1.5, "height": 0.5,
                      "color": (1, 0.9, 0.8), "intensity": 50.0})
    assert L.tobytes() == want.arrays()["lights"].tobytes()
    # the rect-light proxy mesh is mesh 0 (createLight runs before the nodes are processed), glTF meshes follow
    assert len(arr["meshes"]) == 4 and list(arr["instances"]["geom_id"]) == [0, 1, 2, 3]
    # and the whole thing survives the dump format
    dump = os.path.join(tmp_path, "ext.skscene")
    scene_io.save_scene(dump, arr, sc.getCamera(), sc.material_descriptions)
    back = scene_io.load_scene(dump).arrays()
    for k in arr:
        assert flat_bytes(arr[k]) == flat_bytes(back[k]), k


def test_glb_container_and_rejections(tmp_path):
    import struct

    path, _ = make_gltf(tmp_path, embed=True, name="forglb")
    doc = json.load(open(path))
    blob = base64.b64decode(doc["buffers"][0]["uri"].split(",", 1)[1])
    del doc["buffers"][0]["uri"]
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    blob += b"\0" * (-len(blob) % 4)
    glb = os.path.join(tmp_path, "model.glb")
    with open(glb, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + blob)
    a, b = gltf.load_gltf(path).arrays(), gltf.load_gltf(glb).arrays()
    for k in a:
        assert flat_bytes(a[k]) == flat_bytes(b[k]), k
    # non-indexed primitive: the reference asserts (gltfloader.cpp:158)
    doc = json.load(open(path))
    del doc["meshes"][1]["primitives"][0]["indices"]
    bad = os.path.join(tmp_path, "bad.gltf")
    json.dump(doc, open(bad, "w"))
    with pytest.raises(gltf.GltfError):
        gltf.load_gltf(bad)
    doc = json.load(open(path))
    doc["accessors"][3]["count"] = 600  # index accessor running past the buffer
    json.dump(doc, open(bad, "w"))
    with pytest.raises(gltf.GltfError):
        gltf.load_gltf(bad)


def test_corrupt_textures_are_reported_not_fatal(tmp_path, capsys):
    """a corrupt PNG (bad deflate stream, truncated chunk) raised zlib.error / struct.error through the loader and
    aborted the whole scene; now every decoder failure is a PngError and the material keeps its constant colour."""
    import base64
    import zlib

    from strelka_amd import png

    good = tmp_path / "ok.png"
    img = (np.arange(16 * 16 * 4) % 251).astype(np.uint8).reshape(16, 16, 4)
    png.save_png(str(good), img)
    blob = good.read_bytes()
    assert np.array_equal(png.decode_png(blob), img)
    bad_deflate = blob[:60] + bytes(b ^ 0x5A for b in blob[60:80]) + blob[80:]
    truncated = blob[:30]
    for b in (bad_deflate, truncated, blob[:8] + b"\x00\x00"):
        with pytest.raises(png.PngError):
            png.decode_png(b)
    # Sub-filtered rows (the vectorised path) decode to the same pixels as the writer's filter-0 rows
    raw = np.zeros((16, 1 + 64), np.uint8)
    raw[:, 0] = 1
    px = img.reshape(16, 16, 4).astype(np.int32)
    raw[:, 1:] = np.concatenate([px[:, :1], (px[:, 1:] - px[:, :-1]) % 256], 1).reshape(16, 64)
    import struct

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    sub = blob[:8] + chunk(b"IHDR", struct.pack(">IIBBBBB", 16, 16, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw.tobytes())) + chunk(b"IEND", b"")
    assert np.array_equal(png.decode_png(sub), img)
