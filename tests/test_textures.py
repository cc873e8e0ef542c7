"""Textured materials on the CPU side: the oracle's restatement of the reference's texture fetch
(uchar4 / normalized float / linear / wrap: OptixRender.cpp:1191-1264, texture_support_cuda.h:287-313) against an independent
numpy statement of the CUDA programming guide's linear-filtering formula, the material semantics (diffuse texture replaces
the constant, normal map perturbs state.normal), the dump format and the glTF loader carrying textures."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from strelka_amd import gltf, png, scene as S, scene_io, scenes
from tests import orklib
from tests.test_gltf import make_gltf
from tests.test_host_cpp import run_host


def lookup(ork, tex, uv):
    tex = np.ascontiguousarray(tex, np.uint8)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    out = np.zeros((len(uv), 4), np.float32)
    ork.ork_tex_lookup(tex.ctypes.data_as(C.c_void_p), C.c_uint32(tex.shape[1]), C.c_uint32(tex.shape[0]), uv.ctypes.data_as(C.c_void_p),
                       C.c_uint32(len(uv)), out.ctypes.data_as(C.c_void_p))
    return out


def numpy_tex2d(tex, uv):
    """CUDA C Programming Guide, Texture Fetching: normalized coordinates, wrap addressing, linear filtering with 1.8
    fixed-point weights; texel = byte / 255."""
    t = tex.astype(np.float64) / 255.0
    h, w = tex.shape[:2]
    out = np.zeros((len(uv), 4))
    for k, (u, v) in enumerate(np.asarray(uv, np.float32)):
        def axis(c, n):
            c = np.float32(c)
            x = np.float32(np.float32(c - np.floor(c)) * np.float32(n)) - np.float32(0.5)
            i = int(np.floor(x))
            a = np.floor(np.float32(x - np.float32(i)) * 256.0 + 0.5) / 256.0
            return i % n, (i + 1) % n, a
        x0, x1, a = axis(u, w)
        y0, y1, b = axis(v, h)
        out[k] = (1 - a) * (1 - b) * t[y0, x0] + a * (1 - b) * t[y0, x1] + (1 - a) * b * t[y1, x0] + a * b * t[y1, x1]
    return out


def test_texture_fetch_matches_the_cuda_linear_filter_formula(ork):
    rs = np.random.RandomState(5)
    for shape in ((1, 1), (4, 4), (5, 7), (16, 3)):
        tex = rs.randint(0, 256, shape + (4,)).astype(np.uint8)
        uv = np.concatenate([rs.uniform(-3, 3, (400, 2)), [[0, 0], [1, 1], [-1, 0.5], [0.999999, 1e-7], [-1e-8, -1e-8]]]).astype(np.float32)
        got = lookup(ork, tex, uv)
        assert np.allclose(got, numpy_tex2d(tex, uv), atol=2e-6), shape
        # texel centres return the texel exactly; wrap: whole-number shifts of exactly representable coordinates change nothing
        h, w = shape
        ys, xs = np.mgrid[0:h, 0:w]
        centres = np.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], 1).astype(np.float32)
        exact = (w & (w - 1)) == 0 and (h & (h - 1)) == 0
        c = lookup(ork, tex, centres)
        want = tex.reshape(-1, 4).astype(np.float32) / np.float32(255.0)
        assert np.array_equal(c, want) if exact else np.allclose(c, want, atol=1.5 / 256)
        if exact:
            assert np.array_equal(lookup(ork, tex, centres + np.float32([2.0, -3.0])), c)
    # a constant texture is constant everywhere (weights sum to one)
    tex = np.full((8, 8, 4), 77, np.uint8)
    assert np.allclose(lookup(ork, tex, rs.uniform(-2, 2, (200, 2))), 77 / 255.0, atol=1e-6)


def textured_scene(base_tex=None, normal_tex=None, base_color=(0.7, 0.7, 0.7)):
    """a floor quad with UVs 0..2 (so the wrap mode matters) under a rect light, seen from above at an angle"""
    sc = S.Scene()
    bt = sc.addTexture(base_tex) if base_tex is not None else 0
    nt = sc.addTexture(normal_tex) if normal_tex is not None else 0
    sc.addMaterial(S.MAT_PBR, base_color, roughness=0.6, metallic=0.0, base_color_texture=bt, normal_texture=nt)
    p = [(-2, 0, 2), (2, 0, 2), (2, 0, -2), (-2, 0, -2)]
    uv = [(0, 0), (2, 0), (2, 2), (0, 2)]
    idx = [0, 1, 2, 0, 2, 3]
    vb = S.make_vertices([p[i] for i in idx], [(0, 1, 0)] * 6, [uv[i] for i in idx], [(1, 0, 0)] * 6)
    m = sc.createMesh(vb, np.arange(6))
    sc.createInstance(S.INSTANCE_MESH, m, 0, np.eye(4))
    sc.createLight({"type": 0, "useXform": False, "position": (0.0, 3.0, 0.0), "orientation": (-90.0, 0.0, 0.0), "width": 1.5, "height": 1.5,
                    "color": (1.0, 1.0, 1.0), "intensity": 30.0})
    cam = S.Camera(fov=45.0)
    cam.lookAt((0.0, 3.0, 4.5), (0.0, 0.0, 0.0))
    sc.addCamera(cam)
    return sc


def render(sc, w=64, h=48, spp=4, **kw):
    o = orklib.new_context()
    o.set_scene(sc.arrays())
    o.resize(w, h)
    for i in range(spp):
        o.render_subframe(S.frame_params(sc.getCamera(), w, h, subframe_index=i, spp_total=spp, max_depth=3, **kw))
    # debug views bypass the accumulation buffer and go to the output image only (OptixRender.cu:225-247)
    return (o.read_image() if kw.get("debug") else o.read_accum())[..., :3]


def checker(n=8, a=(230, 40, 40, 255), b=(40, 40, 230, 255)):
    t = np.zeros((n, n, 4), np.uint8)
    yy, xx = np.mgrid[0:n, 0:n]
    t[(xx + yy) % 2 == 0] = a
    t[(xx + yy) % 2 == 1] = b
    return t


def test_diffuse_texture_replaces_the_constant_colour():
    grey = np.full((4, 4, 4), 128, np.uint8)
    const = render(textured_scene(base_color=(128 / 255.0,) * 3))
    tex = render(textured_scene(base_tex=grey, base_color=(1.0, 0.0, 0.0)))  # the constant must be ignored
    assert const.max() > 0 and np.allclose(tex, const, rtol=1e-4, atol=1e-6)
    chk = render(textured_scene(base_tex=checker()))
    lit = const.sum(-1) > 0
    assert np.abs(chk - const)[lit].mean() > 1e-3  # the pattern is visible
    r, b = chk[..., 0][lit], chk[..., 2][lit]
    assert (r > 2 * b).any() and (b > 2 * r).any()  # both checker colours show up


def test_normal_map_perturbs_the_shading_normal():
    flat = np.zeros((4, 4, 4), np.uint8)
    flat[...] = (128, 128, 255, 255)
    tilt = np.zeros((4, 4, 4), np.uint8)
    tilt[...] = (255, 128, 128, 255)  # ts = (1, 0.0039, 0.0039): the normal swings onto tangent_u
    base = render(textured_scene(), debug=1)
    f = render(textured_scene(normal_tex=flat), debug=1)
    t = render(textured_scene(normal_tex=tilt), debug=1)
    hit = base.sum(-1) > 0
    assert hit.mean() > 0.3
    n_base = base[hit] * 2 - 1
    assert np.allclose(n_base, (0, 1, 0), atol=2e-3)
    ts = np.array([128, 128, 255]) / 255.0 * 2 - 1
    # frame: tangent_u = +x (vertex tangent), tangent_v = cross(n, tu) = cross(y, x) = -z
    want = ts[0] * np.array([1.0, 0, 0]) + ts[1] * np.array([0, 0, -1.0]) + ts[2] * np.array([0, 1.0, 0])
    want /= np.linalg.norm(want)
    assert np.allclose(f[hit] * 2 - 1, want, atol=2e-3)
    ts = np.array([255, 128, 128]) / 255.0 * 2 - 1
    want = ts[0] * np.array([1.0, 0, 0]) + ts[1] * np.array([0, 0, -1.0]) + ts[2] * np.array([0, 1.0, 0])
    want /= np.linalg.norm(want)
    assert np.allclose(t[hit] * 2 - 1, want, atol=2e-3)
    # and it changes the lighting
    assert np.abs(render(textured_scene(normal_tex=tilt)) - render(textured_scene())).max() > 1e-3


def test_textures_survive_the_dump_format_in_both_directions(tmp_path):
    sc = textured_scene(base_tex=checker(), normal_tex=checker(4))
    arr = sc.arrays()
    p = os.path.join(tmp_path, "tex.skscene")
    scene_io.save_scene(p, arr, sc.getCamera())
    back = scene_io.load_scene(p).arrays()
    assert len(back["textures"]) == 2 and all(np.array_equal(a, b) for a, b in zip(arr["textures"], back["textures"]))
    assert back["materials"].tobytes() == arr["materials"].tobytes()
    out = run_host(tmp_path, "load", p)  # C++ reader -> C++ writer
    assert "load ok" in out
    again = scene_io.load_scene(os.path.join(tmp_path, "resaved.skscene")).arrays()
    assert all(np.array_equal(a, b) for a, b in zip(arr["textures"], again["textures"])) and again["materials"].tobytes() == arr["materials"].tobytes()
    # a material pointing at a texture that is not there is rejected
    bad = dict(arr)
    bad["textures"] = arr["textures"][:1]
    q = os.path.join(tmp_path, "bad.skscene")
    scene_io.save_scene(q, bad, sc.getCamera())
    with pytest.raises(ValueError):
        scene_io.load_scene(q)


def test_gltf_loader_binds_png_textures(tmp_path):
    path, _ = make_gltf(str(tmp_path))
    png.save_png(os.path.join(tmp_path, "albedo.png"), checker(4))
    sc = gltf.load_gltf(path)
    arr = sc.arrays()
    assert len(arr["textures"]) == 1 and np.array_equal(arr["textures"][0], checker(4))
    assert arr["materials"]["base_color_texture"][0] == 1 and arr["materials"]["normal_texture"][0] == 0
    assert arr["materials"]["base_color_texture"][1] == 0
    # without the file the material keeps its constant colour
    os.remove(os.path.join(tmp_path, "albedo.png"))
    arr2 = gltf.load_gltf(path).arrays()
    assert len(arr2["textures"]) == 0 and arr2["materials"]["base_color_texture"][0] == 0


def test_gltf_images_inside_buffers_and_data_uris(tmp_path):
    import base64

    path, _ = make_gltf(str(tmp_path), embed=False, name="bv")
    png.save_png(os.path.join(tmp_path, "albedo.png"), checker(4))
    blob_png = open(os.path.join(tmp_path, "albedo.png"), "rb").read()
    os.remove(os.path.join(tmp_path, "albedo.png"))
    doc = json.load(open(path))
    # (a) data: uri
    doc["images"][0] = {"uri": "data:image/png;base64," + base64.b64encode(blob_png).decode()}
    json.dump(doc, open(path, "w"))
    a = gltf.load_gltf(path).arrays()
    assert len(a["textures"]) == 1 and np.array_equal(a["textures"][0], checker(4)) and a["materials"]["base_color_texture"][0] == 1
    # (b) image bytes appended to the binary buffer, referenced through a bufferView (what .glb files do)
    binp = os.path.join(tmp_path, "bv.bin")
    data = open(binp, "rb").read()
    off = len(data) + (-len(data) % 4)
    open(binp, "wb").write(data + b"\0" * (off - len(data)) + blob_png)
    doc["buffers"][0]["byteLength"] = off + len(blob_png)
    doc["bufferViews"].append({"buffer": 0, "byteOffset": off, "byteLength": len(blob_png)})
    doc["images"][0] = {"bufferView": len(doc["bufferViews"]) - 1, "mimeType": "image/png"}
    json.dump(doc, open(path, "w"))
    b = gltf.load_gltf(path).arrays()
    assert len(b["textures"]) == 1 and np.array_equal(b["textures"][0], checker(4)) and b["materials"]["base_color_texture"][0] == 1


def test_png_round_trip_and_screenshot_orientation(tmp_path):
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, (9, 14, 4)).astype(np.uint8)
    p = os.path.join(tmp_path, "a.png")
    png.save_png(p, img)
    assert np.array_equal(png.load_png(p), img)
    f = rs.rand(6, 5, 4).astype(np.float32) * 1.4 - 0.2  # out-of-range values are clamped
    png.save_png(p, f, flipped=True)  # hdRunner: storage.flipped = true (main.cpp:430)
    back = png.load_png(p)
    want = (np.clip(f, 0, 1).astype(np.float64) * 255.0 + 0.5).astype(np.uint8)[::-1]
    assert np.array_equal(back, want)
    with pytest.raises(png.PngError):
        open(p, "wb").write(b"not a png")
        png.load_png(p)


def test_baseline_jpeg_reader(tmp_path):
    """JPEG textures (what stbi_load decodes for most glTF assets): the reader against images written by the module's own
    baseline encoder -- 4:4:4 and 4:2:0, with and without restart intervals, sizes that are not multiples of the MCU -- within
    the quantisation error; progressive files and non-JPEG data are rejected."""
    from strelka_amd import jpeg

    rs = np.random.RandomState(4)
    yy, xx = np.mgrid[0:45, 0:61]
    img = np.stack([127 + 100 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 127 + 90 * np.cos(xx / 11.0), 60 + 3 * yy + rs.randint(0, 8, (45, 61))],
                   -1).clip(0, 255).astype(np.uint8)
    p = os.path.join(tmp_path, "t.jpg")
    for sub, ri, tol in ((False, 0, 2.5), (False, 5, 2.5), (True, 0, 5.0), (True, 2, 5.0)):
        jpeg.save_jpeg(p, img, quality_scale=0.25, subsample=sub, restart_interval=ri)
        out = jpeg.load_jpeg(p)
        assert out.shape == (45, 61, 4) and (out[..., 3] == 255).all()
        err = np.abs(out[..., :3].astype(int) - img.astype(int))
        assert err.mean() < tol and err.max() < 40, (sub, ri, err.mean(), err.max())
    flat = np.full((16, 16, 3), 200, np.uint8)
    jpeg.save_jpeg(p, flat, quality_scale=0.1)
    assert np.abs(jpeg.load_jpeg(p)[..., :3].astype(int) - 200).max() <= 1  # DC only: exact up to rounding
    blob = bytearray(open(p, "rb").read())
    blob[blob.index(b"\xff\xc0") + 1] = 0xC2  # pretend progressive
    with pytest.raises(jpeg.JpegError):
        jpeg.decode_jpeg(bytes(blob))
    with pytest.raises(jpeg.JpegError):
        jpeg.decode_jpeg(b"\x89PNG....")
    # the glTF loader picks the decoder by the file's magic bytes
    path, _ = make_gltf(str(tmp_path), name="jp")
    jpeg.save_jpeg(os.path.join(tmp_path, "albedo.png"), checker(16)[..., :3], quality_scale=0.1)  # JPEG content under the model's uri
    arr = gltf.load_gltf(path).arrays()
    assert len(arr["textures"]) == 1 and arr["textures"][0].shape == (16, 16, 4)
    assert np.abs(arr["textures"][0][..., :3].astype(int) - checker(16)[..., :3].astype(int)).mean() < 12
