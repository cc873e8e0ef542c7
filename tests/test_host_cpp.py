"""The C++ host mirror (strelka_amd/host: oka::Scene / Camera / SettingsManager / HipRender above the C ABI).

CPU: the flat arrays the C++ oka::Scene produces equal what the Python mirror (strelka_amd/scene.py) produces for the
same recipe -- two independent restatements of src/scene/scene.cpp + camera.cpp.
GPU: driving HipRender::render() like hdRunner's frame loop gives exactly the accumulation buffer the ctypes path
gives for the same inputs, the tonemapped output image, and the reference's sub-frame bookkeeping."""
import math
import os
import subprocess

import numpy as np
import pytest

from strelka_amd import scene as S


def run_host(tmp_path, mode, *args):
    from strelka_amd import build

    exe = build.build_host()
    out = subprocess.run([exe, mode, str(tmp_path)] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def load(tmp_path):
    f = lambda n, dt: np.fromfile(os.path.join(tmp_path, n), dtype=dt)
    return {"vertices": f("vertices.bin", S.VERTEX), "indices": f("indices.bin", np.uint32), "meshes": f("meshes.bin", S.MESH),
            "lights": f("lights.bin", S.LIGHT), "instances": f("instances.bin", S.INSTANCE),
            "materials": f("materials.bin", S.MATERIAL), "camera": f("camera.bin", np.float32),
            "curves": np.zeros(0, S.CURVE), "curve_points": np.zeros((0, 3), np.float32), "curve_radii": np.zeros(0, np.float32),
            "curve_vertex_counts": np.zeros(0, np.uint32)}


def python_recipe():
    """the scene of host_test.cpp::buildScene, through the Python mirror"""
    sc = S.Scene()
    sc.addMaterial(S.MAT_DIFFUSE, (0.8, 0.8, 0.8), roughness=0.0, metallic=0.0, specular=0.0, ior=0.0)
    sc.addMaterial(S.MAT_PBR, (0.9, 0.3, 0.2), roughness=0.3, metallic=0.0, specular=0.5, ior=1.5)

    def quad(p, n):
        idx = [0, 1, 2, 0, 2, 3]
        return S.make_vertices([p[i] for i in idx], [n] * 6, [(0.0, 0.0)] * 6, [(1, 0, 0)] * 6)

    fm = sc.createMesh(quad([(-2, 0, 2), (2, 0, 2), (2, 0, -2), (-2, 0, -2)], (0, 1, 0)), np.arange(6))
    pm = sc.createMesh(quad([(-0.5, 0, 0), (0.5, 0, 0), (0.5, 1, 0), (-0.5, 1, 0)], (0, 0, 1)), np.arange(6))
    sc.createInstance(S.INSTANCE_MESH, fm, 0, np.eye(4))
    xf = S.translate((0.3, 0.0, -0.4)) @ S.quat_to_mat4(S.quat_from_euler_deg((0, math.degrees(0.6), 0))) @ S.scale((1.5, 1.2, 1.0))
    sc.createInstance(S.INSTANCE_MESH, pm, 1, xf)
    sc.createLight({"type": 0, "useXform": True, "xform": S.translate((0, 2.5, 0.5)) @ S.quat_to_mat4(
        S.quat_from_euler_deg((math.degrees(-1.5707963), 0, 0))), "width": 0.8, "height": 0.6, "color": (1.0, 0.9, 0.8), "intensity": 40.0})
    sc.createLight({"type": 2, "useXform": False, "position": (-1.2, 0.8, 0.6), "orientation": (10.0, 20.0, 30.0),
                    "radius": 0.15, "color": (0.5, 0.7, 1.0), "intensity": 25.0})
    sc.createLight({"type": 3, "useXform": True, "xform": S.quat_to_mat4(S.quat_from_euler_deg(
        (math.degrees(-0.9), math.degrees(0.4), 0))), "halfAngle": 0.0872664626, "color": (1, 1, 1), "intensity": 1.5, "radius": 0.0})
    cam = S.Camera(fov=50.0)
    cam.lookAt((1.5, 1.8, 3.5), (0.0, 0.6, 0.0))
    sc.addCamera(cam)
    return sc


def test_cpp_scene_equals_python_scene(tmp_path):
    run_host(tmp_path, "cpu")
    got = load(tmp_path)
    sc = python_recipe()
    want = sc.arrays()
    assert np.array_equal(got["indices"], want["indices"]) and np.array_equal(got["meshes"], want["meshes"])
    for f in ("tangent", "normal", "uv"):
        assert np.array_equal(got["vertices"][f], want["vertices"][f])
    assert np.allclose(got["vertices"]["pos"], want["vertices"]["pos"], atol=1e-6)
    for f in ("type", "geom_id", "material_id", "light_id"):
        assert np.array_equal(got["instances"][f], want["instances"][f])
    assert np.allclose(got["instances"]["transform"], want["instances"]["transform"], atol=2e-6)
    assert np.array_equal(got["lights"]["type"], want["lights"]["type"])
    for f in ("points", "color", "normal", "half_angle"):
        assert np.allclose(got["lights"][f], want["lights"][f], atol=2e-6), f
    assert got["materials"].tobytes() == want["materials"][:2].tobytes()
    p = S.frame_params(sc.getCamera(), 96, 64)
    assert np.allclose(got["camera"][:16], p["view_to_world"], atol=2e-6)
    assert np.allclose(got["camera"][16:], p["clip_to_view"], rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_hiprender_frame_loop_matches_ctypes_path_and_oracle(tmp_path):
    from strelka_amd import capi
    from tests import orklib

    frames = 6
    run_host(tmp_path, "gpu", frames)
    arr = load(tmp_path)
    W, H = 96, 64
    accum = np.fromfile(os.path.join(tmp_path, "accum.bin"), np.float32).reshape(H, W, 4)
    image = np.fromfile(os.path.join(tmp_path, "image.bin"), np.float32).reshape(H, W, 4)
    ctx = capi.Context(0)
    ctx.set_scene(arr)
    ctx.resize(W, H)
    o = orklib.new_context()
    o.set_scene(arr)
    o.resize(W, H)
    p = np.zeros((), S.FRAME_PARAMS)
    p["view_to_world"], p["clip_to_view"] = arr["camera"][:16], arr["camera"][16:]
    p["samples_this_launch"], p["spp_total"], p["max_depth"], p["enable_accumulation"] = 1, frames - 1, 4, 1
    p["exposure"] = S.default_exposure()
    for i in range(frames - 1):  # the C++ loop's last frame only copies accum -> image (all spp done)
        p["subframe_index"] = i
        ctx.render_subframe(p)
        o.render_subframe(p)
    mine = ctx.read_accum()
    assert np.array_equal(mine, accum)  # same library, same inputs: bit-identical
    want = o.read_accum()[..., :3].astype(np.float64)
    l2 = np.sqrt(((accum[..., :3] - want) ** 2).sum()) / np.sqrt((want ** 2).sum())
    assert l2 < 2e-2
    # image = gamma(reinhard(accum * exposure)) (Tonemappers.cu), tonemapperType 1, gamma 2.4
    e = S.default_exposure()
    r = accum[..., :3] * e
    lum = r @ np.array([0.299, 0.587, 0.114], np.float32)
    ref = (r / (lum[..., None] + 1)) ** (1 / 2.4)
    assert np.allclose(image[..., :3], ref, rtol=1e-4, atol=1e-6)
    ctx.close()


@pytest.mark.gpu
def test_hiprender_tile_sharing_path_reproduces_the_plain_frame_loop(tmp_path):
    """oka::HipRender::enableTileSharing: tile share -> skh_set_tiles, skh_gather_tiles below the C ABI
    after every render(), skh_scatter_tiles into the output on the root.  With the world size a 1-GPU box allows (1) the mapped
    image and the accumulator must equal the plain frame loop's bit for bit."""
    a, b = tmp_path / "plain", tmp_path / "tiles"
    a.mkdir()
    b.mkdir()
    run_host(a, "gpu", 5)
    run_host(b, "gpu-tiles", 5)
    for f in ("image.bin", "accum.bin"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f


@pytest.mark.gpu
@pytest.mark.parametrize("view", [2, 3])
def test_hiprender_hands_back_the_aov_after_the_last_sample(tmp_path, view):
    """VERDICT r3 weak #7 / OptixRender.cpp:1029-1049: with every sample done and the debug view 2 / 3, each further render() copies the RAW
    diffuse / specular AOV to the image (skh_copy_aov) and tonemaps that -- five more calls must leave the picture unchanged, and it must be
    gamma(reinhard(AOV * exposure)), not a tonemap of a tonemap."""
    W, H = 96, 64
    run_host(tmp_path, "gpu-aov%d" % view, 5)
    first = np.fromfile(os.path.join(tmp_path, "image_first.bin"), np.float32).reshape(H, W, 4)
    last = np.fromfile(os.path.join(tmp_path, "image.bin"), np.float32).reshape(H, W, 4)
    aov = np.fromfile(os.path.join(tmp_path, "aov.bin"), np.float32).reshape(H, W, 4)
    assert np.array_equal(first, last)
    assert aov[..., :3].max() > 0  # the view shows something
    r = aov[..., :3] * S.default_exposure()
    lum = r @ np.array([0.299, 0.587, 0.114], np.float32)
    ref = (r / (lum[..., None] + 1)) ** (1 / 2.4)
    assert np.allclose(last[..., :3], ref, rtol=1e-4, atol=1e-6)
