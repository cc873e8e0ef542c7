"""Render-level checks of the CPU oracle (the restated __raygen__rg loop): determinism, sub-frame bookkeeping,
estimator consistency.  Small images so the CPU suite stays fast."""
import numpy as np

from strelka_amd import scene as S
from strelka_amd import scenes
from tests import orklib


def render(sc, w, h, spp, depth=4, **kw):
    o = orklib.new_context()
    o.set_scene(sc.arrays())
    o.resize(w, h)
    for i in range(spp):
        o.render_subframe(S.frame_params(sc.getCamera(), w, h, subframe_index=i, spp_total=spp, max_depth=depth, **kw))
    return o


def test_deterministic_and_row_ranges_compose():
    sc = scenes.cornell_box()
    a = render(sc, 40, 32, 2).read_accum()
    b = render(sc, 40, 32, 2).read_accum()
    assert np.array_equal(a, b)
    o = orklib.new_context()
    o.set_scene(sc.arrays())
    o.resize(40, 32)
    for i in range(2):
        p = S.frame_params(sc.getCamera(), 40, 32, subframe_index=i, spp_total=2)
        o.render_subframe(p, rows=(0, 10))
        o.render_subframe(p, rows=(10, 32))
    assert np.array_equal(o.read_accum(), a)  # pixels are independent: any tiling gives the same image


def test_direct_light_is_visible_and_shadows_exist():
    sc = scenes.cornell_box()
    img = render(sc, 48, 48, 8).read_accum()[..., :3]
    lum = img.sum(axis=-1)
    y, x = np.unravel_index(np.argmax(lum), lum.shape)
    assert lum[y, x] > 5.0 and y > 32 and 14 < x < 34  # the ceiling light seen directly (row 0 = bottom of the image)
    assert 0.01 < img[6:10, 4:12].mean() < 2.0  # lit floor
    assert np.isfinite(img).all() and (img >= 0).all()
    # red wall on the left, green on the right
    left, right = img[20:30, 1:4].mean(axis=(0, 1)), img[20:30, 44:47].mean(axis=(0, 1))
    assert left[0] > 2 * left[1] and right[1] > 1.5 * right[0]


def test_rect_light_sampling_methods_agree_in_the_mean():
    """rectLightSamplingMethod 0 (uniform area) and 1 (spherical rectangle) are two estimators of the same integral."""
    sc = scenes.cornell_box()
    a = render(sc, 24, 24, 48, depth=2, rect_light_sampling_method=0).read_accum()[..., :3]
    b = render(sc, 24, 24, 48, depth=2, rect_light_sampling_method=1).read_accum()[..., :3]
    ma, mb = a[2:12].mean(), b[2:12].mean()  # floor region
    assert abs(ma - mb) / ma < 0.08


def test_russian_roulette_only_beyond_depth_3():
    """depth 4: RR never fires (depth in 0..3 at test time, OptixRender.cu:134); depth 6 adds energy, never removes it on average."""
    sc = scenes.cornell_box()
    d4 = render(sc, 24, 24, 24, depth=4).read_accum()[..., :3].mean()
    d6 = render(sc, 24, 24, 24, depth=6).read_accum()[..., :3].mean()
    assert d6 > d4 * 0.98


def test_multi_sample_launch_and_aov_counters():
    sc = scenes.kitchen_standin(seed=2, n_meshes=6, n_instances=61, tri_lo=40, tri_hi=200)
    o = orklib.new_context()
    o.set_scene(sc.arrays())
    o.resize(32, 24)
    o.render_subframe(S.frame_params(sc.getCamera(), 32, 24, subframe_index=0, samples_this_launch=3, spp_total=6))
    o.render_subframe(S.frame_params(sc.getCamera(), 32, 24, subframe_index=3, samples_this_launch=3, spp_total=6))
    img = o.read_accum()
    assert np.isfinite(img).all() and img[..., :3].max() > 0
    d, s = o.read_aov(0), o.read_aov(1)
    assert np.isfinite(d).all() and np.isfinite(s).all() and d[..., :3].max() > 0
    st = o.stats()
    assert st["rays_radiance"] >= 32 * 24 * 6 and st["rays_shadow"] > 0


def test_debug_normals_view():
    sc = scenes.cornell_box()
    o = orklib.new_context()
    o.set_scene(sc.arrays())
    o.resize(16, 16)
    o.render_subframe(S.frame_params(sc.getCamera(), 16, 16, debug=1, enable_accumulation=0))
    img = o.read_image()[..., :3]
    # back wall normal +z -> (0.5, 0.5, 1.0) after (n + 1) / 2, within the 10-bit packing error
    assert np.allclose(img[12, 8], (0.5, 0.5, 1.0), atol=3e-3)


def test_direct_lighting_of_a_floor_under_a_rect_light_has_its_closed_form():
    """End-to-end known answer for the restated integrator (raygen -> hit reconstruction -> Lambert sample / evaluate -> NEE with the light's pdf ->
    light hit with MIS -> per-launch mean), independent of any reference arithmetic: a large diffuse floor (albedo rho) under a rectangular light
    parallel to it.  By the reference's own estimator definitions (closest_hit.cu:294-306,586-591; __closesthit__light: emission x cos at the
    LIGHT; NEE: Li x cos at the SURFACE x the BSDF's own cosine) both MIS branches estimate, for parallel planes where the two cosines are equal,
        L_o = rho L / pi  x  Integral over the rectangle of cos^2(theta) d(omega)  =  rho L / pi  x  Integral H^3 / r^5 dA,
    evaluated here by fp64 quadrature.  max_depth 2 (direct light only: the flat floor cannot light itself), one launch of many samples (a linear
    mean: the sub-frame accumulator is a tonemapped lerp), both rect sampling methods.  Monte-Carlo bar 1.5 % (65 536 samples per method;
    measured 0.3751 and 0.3752 against 0.3771: -0.5 %, of which ~0.2 % is the pixels' footprint around the origin)."""
    sc, want = floor_under_rect_light()
    for method in (0, 1):
        o = orklib.new_context()
        o.set_scene(sc.arrays())
        o.resize(8, 8)
        spp = 1024
        o.render_subframe(S.frame_params(sc.getCamera(), 8, 8, subframe_index=0, samples_this_launch=spp, spp_total=spp, max_depth=2,
                                         rect_light_sampling_method=method))
        img = o.read_accum()[..., :3]
        got = float(img.mean())
        assert np.allclose(img.mean(axis=(0, 1)), got, rtol=1e-6)  # grey in, grey out
        assert abs(got - want) <= 0.015 * want, (method, got, want)


def floor_under_rect_light():
    """the scene of the closed-form test and its fp64 answer (also rendered by the HIP path: tests/test_gpu_parity.py)"""
    import math

    rho, L, H, a, b = 0.5, 10.0, 1.5, 1.0, 0.6
    sc = S.Scene()
    grey = sc.addMaterial(S.MAT_DIFFUSE, (rho, rho, rho))
    vb, ib = S.deindex(np.array([(-20, 0, 20), (20, 0, 20), (20, 0, -20), (-20, 0, -20)], np.float32), np.array([(0, 1, 2), (0, 2, 3)]))  # normal +y
    sc.createInstance(S.INSTANCE_MESH, sc.createMesh(vb, ib), grey, np.eye(4))
    xf = S.translate((0.0, H, 0.0)) @ S.rotate((1, 0, 0), math.radians(-90))  # local -Z (the emitting side, Lights.h:54-62) -> world -Y
    sc.createLight({"type": 0, "xform": xf, "useXform": True, "width": a, "height": b, "color": (L, L, L), "intensity": 1.0})
    cam = S.Camera(fov=1.5)
    cam.lookAt((3.0, 1.0, 0.4), (0.0, 0.0, 0.0))
    sc.addCamera(cam)
    # fp64 quadrature of H^3 / r^5 over the rectangle as seen from the origin (the 8 x 8 pixels see the floor within 4 cm of it: 0.2 %)
    n = 1200
    xs = (np.arange(n) + 0.5) / n * a - a / 2
    zs = (np.arange(n) + 0.5) / n * b - b / 2
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    r2 = X * X + Z * Z + H * H
    return sc, rho * L / math.pi * float((H ** 3 / r2 ** 2.5).sum() * (a / n) * (b / n))


def _l2_and_outliers(x, y):
    x, y = x[..., :3].astype(np.float64), y[..., :3].astype(np.float64)
    l2 = np.sqrt(((x - y) ** 2).sum()) / np.sqrt((y ** 2).sum())
    rel = np.abs(x - y).max(-1) / np.maximum(y.max(-1), 1e-6)
    return l2, float((rel > 1e-4).mean())


def test_shared_libm_against_a_glibc_build_of_the_checker():
    """The default checker compiles the product's skh_libm.h (so that GPU and checker agree bit for bit); that makes an error in one of its
    polynomials invisible to every GPU-vs-checker comparison.  Second opinion: the SAME checker source built with glibc's transcendentals
    (oracle/Makefile: liboracle_glibc.so, -DORK_LIBM_GLIBC) renders the same scenes -- Lambert (cosine sampling: sin / cos), the mixed-material
    kitchen (GGX / glass), every light type (sphere: sin / cos / acos; distant: cos; spherical rectangle: acos chains) and hair (Chiang: exp /
    log / atan2 / asin / sinh) -- and the images agree at round 4's bar, relative L2 <= 2e-5, with <= 0.5 % of the pixels further than 1e-4 apart
    (a last-ulp difference in a direction may move single paths across an edge; measured: L2 2e-8 ... 1.3e-6, 0 ... 0.11 % of the pixels)."""
    a, b = orklib.load(), orklib.load_glibc()

    def render_with(lib, sc, w, h, spp, depth, **kw):
        o = orklib.Oracle(lib)
        o.set_scene(sc.arrays())
        o.resize(w, h)
        for i in range(spp):
            o.render_subframe(S.frame_params(sc.getCamera(), w, h, subframe_index=i, spp_total=spp, max_depth=depth, **kw))
        return o.read_accum()

    differ = 0
    for sc, w, h, spp, depth, kw in [(scenes.cornell_box(), 64, 64, 8, 4, {}),
                                     (scenes.kitchen_standin(seed=7, n_meshes=12, n_instances=60, tri_lo=100, tri_hi=1500), 96, 64, 4, 5, {}),
                                     (scenes.hair_standin(n_strands=2000), 96, 64, 4, 3, {}),
                                     (scenes.light_zoo(), 64, 64, 4, 4, {}),
                                     (scenes.light_zoo(), 64, 64, 4, 4, {"rect_light_sampling_method": 1})]:
        x, y = render_with(a, sc, w, h, spp, depth, **kw), render_with(b, sc, w, h, spp, depth, **kw)
        l2, off = _l2_and_outliers(x, y)
        assert l2 <= 2e-5 and off <= 0.005, (l2, off)
        differ += not np.array_equal(x, y)
    assert differ >= 3  # the two builds really are different arithmetic (else this test compares a library with itself)
