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
