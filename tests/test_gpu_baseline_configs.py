"""The remaining BASELINE.json configurations as parity cases (the bench line is C3, tests/test_gpu_fullsize.py):
  C1  coffeemaker stand-in, 512x512, 2 bounces, 16 spp   -- the whole configuration on both sides
  C2  Cornell box, 1024x1024, 4 bounces, 256 spp          -- oracle on 2 spp at full resolution; 256 spp through properties
  C4  kitchen stand-in, 3840x2160, 6 bounces, 8 tile sets -- Russian roulette (depth > 3) and the 8-rank tile split at 4K (16 spp);
      the whole 256-spp frame once through properties + an oracle band of its 64-sub-frame prefix
  C5  hair stand-in, 1920x1080, 3 bounces                 -- full-resolution properties (hit parity is in test_gpu_fullsize) + a 32-row
      band through the hair, sub-frames 0..3 of 1024, against the oracle: image equal, ray counts equal (round 6)
Images are compared with tests/test_gpu_parity.py::_image_equal (bit for bit since round 5); ray counts and tile-sharded images are exact."""
import numpy as np
import pytest

from strelka_amd import scene as S, scenes, tiles
from tests.tilehelp import detile_numpy

pytestmark = pytest.mark.gpu


def _oracle_frame(arr, cam, W, H, spp, depth, first=0, total=None):
    from tests import orklib

    o = orklib.new_context()
    o.set_scene(arr)
    o.resize(W, H)
    for i in range(spp):
        o.render_subframe(S.frame_params(cam, W, H, subframe_index=first + i, samples_this_launch=1, spp_total=total or spp, max_depth=depth))
    return o.read_accum(), o.stats()


def _gpu_frame(arr, cam, W, H, spp, depth, options=None, tile_xy=None, total=None):
    import torch
    from strelka_amd import capi

    ctx = capi.Context(0)
    for k, v in (options or {}).items():
        ctx.set_option(k, v)
    ctx.set_scene(arr)
    ctx.set_tiles(32, tile_xy)
    ctx.resize(W, H)
    ctx.render_subframes(S.frame_params(cam, W, H, subframe_index=0, samples_this_launch=1, spp_total=total or spp, max_depth=depth), spp, None)
    st = ctx.stats()
    if tile_xy is None:
        out = ctx.read_accum()
    else:
        buf = torch.zeros((len(tile_xy), 32 * 32, 4), dtype=torch.float32, device="cuda")
        ctx.copy_accum_tiles(buf.data_ptr())
        out = buf.cpu().numpy()
    ctx.close()
    return out, st


def test_c1_coffeemaker_512_whole_configuration_matches_oracle():
    from tests.test_gpu_parity import _image_equal

    sc = scenes.coffeemaker_standin()
    arr = sc.arrays()
    assert 40_000 <= len(arr["indices"]) // 3 <= 70_000
    W = H = 512
    want, so = _oracle_frame(arr, sc.getCamera(), W, H, 16, 2)
    got, st = _gpu_frame(arr, sc.getCamera(), W, H, 16, 2)
    _image_equal(got, want)
    assert st["rays_radiance"] == so["rays_radiance"] and W * H * 16 <= st["rays_radiance"] <= W * H * 16 * 2
    assert want[..., :3].max() > 0


def test_c2_cornell_1024_oracle_on_two_samples_and_256_spp_properties():
    from tests.test_gpu_parity import _image_equal

    sc = scenes.cornell_box()
    arr = sc.arrays()
    W = H = 1024
    want, so = _oracle_frame(arr, sc.getCamera(), W, H, 2, 4, total=256)
    got, st = _gpu_frame(arr, sc.getCamera(), W, H, 2, 4, total=256)
    _image_equal(got, want)
    assert st["rays_radiance"] == so["rays_radiance"]
    # the whole 256-spp frame: the same bits whatever the pass size, every pixel finite and non-negative
    full, stf = _gpu_frame(arr, sc.getCamera(), W, H, 256, 4)
    again, _ = _gpu_frame(arr, sc.getCamera(), W, H, 256, 4, options={"subframe_batch": 8})
    assert full.tobytes() == again.tobytes()
    assert np.isfinite(full).all() and 0.0 <= full[..., :3].min() and full[..., :3].max() > 1.0  # the emitter is in view
    assert W * H * 256 <= stf["rays_radiance"] <= W * H * 256 * 4
    # converged enough to be compared with the 2-sample image in the mean
    assert abs(full[..., :3].mean() - got[..., :3].mean()) < 0.1 * full[..., :3].mean()


def test_c4_kitchen_4k_depth6_roulette_and_eight_rank_tiles():
    from tests.test_gpu_parity import _image_equal

    sc = scenes.kitchen_standin()
    arr = sc.arrays()
    W, H, DEPTH, SPP = 3840, 2160, 6, 16  # 16 of the configuration's 256 spp (was 2)
    base, st = _gpu_frame(arr, sc.getCamera(), W, H, SPP, DEPTH, total=256)
    assert np.isfinite(base).all() and base[..., :3].max() > 0
    # depth 6 goes past the Russian-roulette threshold (OptixRender.cu:131-146): some paths are longer than 4 segments
    _, st4 = _gpu_frame(arr, sc.getCamera(), W, H, SPP, 4, total=256)
    assert st["rays_radiance"] > st4["rays_radiance"]
    # the 8-rank tile split of C4: ranks 0, 3 and 7 reproduce their pixels of the full frame bit for bit
    for rank in (0, 3, 7):
        t = tiles.assign_tiles(W, H, 32, 8, rank)
        tacc, _ = _gpu_frame(arr, sc.getCamera(), W, H, SPP, DEPTH, tile_xy=t, total=256)
        part = detile_numpy(tacc, t, 32, W, H)
        mask = detile_numpy(np.ones_like(tacc), t, 32, W, H)[..., 0] > 0
        assert mask.mean() == pytest.approx(1 / 8, abs=0.01)
        assert part[mask].tobytes() == base[mask].tobytes()
    # a band of the 4K frame against the oracle (32 rows of all 16 sub-frames on the host cores)
    from tests import orklib

    o = orklib.new_context()
    o.set_scene(arr)
    o.resize(W, H)
    r0, r1 = 1024, 1056
    for i in range(SPP):
        o.render_subframe(S.frame_params(sc.getCamera(), W, H, subframe_index=i, samples_this_launch=1, spp_total=256, max_depth=DEPTH), rows=(r0, r1))
    # (round 4 held this band at L2 <= 2e-3 / 0.06 % of the pixels: ~1.3e-5 of the paths were flipped across an edge by libm differences;
    # with the shared skh_libm.h the band is equal bit for bit)
    _image_equal(base[r0:r1], o.read_accum()[r0:r1])


def test_c4_kitchen_4k_full_256_spp_frame_properties():
    """The whole C4 frame once on one GPU (the suite ran 16 of its 256 spp): 3840x2160, depth 6, 256 sub-frames of
    1 spp = 2.1 G paths, Russian roulette active.  Properties as for C5's full frame -- every pixel finite and non-negative, ray counts
    inside their bounds and per sample within 1 % of a 16-spp run's, the uint16 AOV counters intact (diffuse + specular first events
    <= 256 per pixel), no traversal-stack overflow, the converged image smoother than the 16-spp one with the same mean -- plus a 32-row
    band of a 64-sub-frame prefix of the SAME frame (spp_total 256) against the oracle."""
    from strelka_amd import capi
    from tests import orklib
    from tests.test_gpu_parity import _image_equal

    sc = scenes.kitchen_standin()
    arr = sc.arrays()
    W, H, DEPTH, SPP = 3840, 2160, 6, 256
    few, stf = _gpu_frame(arr, sc.getCamera(), W, H, 16, DEPTH, total=SPP)
    ctx = capi.Context(0)
    ctx.set_scene(arr)
    ctx.resize(W, H)
    p = S.frame_params(sc.getCamera(), W, H, subframe_index=0, samples_this_launch=1, spp_total=SPP, max_depth=DEPTH)
    ctx.render_subframes(p, 64, None)
    band64 = ctx.read_accum()[1024:1056].copy()  # the frame after its first 64 sub-frames
    p["subframe_index"] = 64
    ctx.render_subframes(p, SPP - 64, None)  # ... continued to the end
    st = ctx.stats()
    full, dif, spec = ctx.read_accum(), ctx.read_aov(0), ctx.read_aov(1)
    ctx.close()
    assert st["stack_overflows"] == 0
    for img in (full, dif, spec):
        assert np.isfinite(img).all() and img[..., :3].min() >= 0.0
    assert W * H * SPP <= st["rays_radiance"] <= W * H * SPP * DEPTH and st["rays_shadow"] <= st["rays_radiance"]
    assert abs(st["rays_radiance"] / SPP - stf["rays_radiance"] / 16) < 0.01 * stf["rays_radiance"] / 16
    assert abs(full[..., :3].mean() - few[..., :3].mean()) < 0.05 * few[..., :3].mean()

    def rough(img):
        g = img[..., 1]
        return np.abs(4 * g[1:-1, 1:-1] - g[:-2, 1:-1] - g[2:, 1:-1] - g[1:-1, :-2] - g[1:-1, 2:]).mean()

    assert rough(full) < 0.6 * rough(few)
    o = orklib.new_context()
    o.set_scene(arr)
    o.resize(W, H)
    for i in range(64):
        o.render_subframe(S.frame_params(sc.getCamera(), W, H, subframe_index=i, samples_this_launch=1, spp_total=SPP, max_depth=DEPTH), rows=(1024, 1056))
    _image_equal(band64, o.read_accum()[1024:1056])


def test_c5_hair_1080p_depth3_properties():
    sc = scenes.hair_standin()
    arr = sc.arrays()
    W, H = 1920, 1080
    a, st = _gpu_frame(arr, sc.getCamera(), W, H, 4, 3, total=1024)
    b, st2 = _gpu_frame(arr, sc.getCamera(), W, H, 4, 3, total=1024, options={"subframe_batch": 1, "curve_split": 3})
    assert np.isfinite(a).all() and a[..., :3].max() > 0
    assert a.tobytes() == b.tobytes() and st["rays_radiance"] == st2["rays_radiance"] and st["rays_shadow"] == st2["rays_shadow"]
    assert W * H * 4 <= st["rays_radiance"] <= W * H * 4 * 3
    t = tiles.assign_tiles(W, H, 32, 8, 5)
    tacc, _ = _gpu_frame(arr, sc.getCamera(), W, H, 4, 3, tile_xy=t, total=1024)
    part = detile_numpy(tacc, t, 32, W, H)
    mask = detile_numpy(np.ones_like(tacc), t, 32, W, H)[..., 0] > 0
    assert part[mask].tobytes() == a[mask].tobytes()


def test_c5_hair_1080p_band_matches_oracle():
    """C5 at its own size against the checker (the 1080p C5 tests were property-only while C3 and C4 each had an
    oracle band): 1920x1080, depth 3, sub-frames 0..3 of spp_total 1024, the 32 rows 512..543 -- through the middle of the hair --
    `_image_equal` (zero differing pixels), and the same band rendered as a tile set (the row of 60 32x32 tiles) has the checker's
    radiance- and shadow-ray counts exactly (the GPU skips zero-contribution shadow rays: <=).  Exercises the world-only curve kernel,
    its light-proxy rule, the 16-byte curve hit record and the hair build of k_shade at the configuration's resolution."""
    from tests import orklib
    from tests.test_gpu_parity import _image_equal

    sc = scenes.hair_standin()
    arr = sc.arrays()
    W, H, DEPTH, SPP, TOTAL = 1920, 1080, 3, 4, 1024
    r0, r1 = 512, 544
    full, _ = _gpu_frame(arr, sc.getCamera(), W, H, SPP, DEPTH, total=TOTAL)
    o = orklib.new_context()
    o.set_scene(arr)
    o.resize(W, H)
    for i in range(SPP):
        o.render_subframe(S.frame_params(sc.getCamera(), W, H, subframe_index=i, samples_this_launch=1, spp_total=TOTAL, max_depth=DEPTH), rows=(r0, r1))
    want = o.read_accum()[r0:r1]
    so = o.stats()
    assert (want[..., :3].sum(-1) > 0).mean() > 0.2  # the band is not background
    _image_equal(full[r0:r1], want)
    grid = tiles.tile_grid(W, H, 32)
    band_tiles = np.ascontiguousarray(grid[grid[:, 1] == r0])
    assert len(band_tiles) == W // 32
    tacc, st = _gpu_frame(arr, sc.getCamera(), W, H, SPP, DEPTH, tile_xy=band_tiles, total=TOTAL)
    part = detile_numpy(tacc, band_tiles, 32, W, H)
    assert part[r0:r1].tobytes() == full[r0:r1].tobytes()
    assert st["rays_radiance"] == so["rays_radiance"], (st["rays_radiance"], so["rays_radiance"])
    assert 0.9 * so["rays_shadow"] <= st["rays_shadow"] <= so["rays_shadow"]


def test_c5_hair_full_1024_spp_frame_properties():
    """The whole C5 configuration once (the suite ran 4 of its 1024 spp): 1920x1080, depth 3, 1024 sub-frames of 1 spp
    = 2.1 G paths with the Chiang hair BSDF.  Properties: every pixel finite and non-negative, ray counts inside their bounds and
    exactly 256x those of a 4-spp run's per-sample average within 1 %, the converged image's mean near the 4-spp image's, the
    uint16 AOV counters intact (diffuse + specular first events <= 1024 per pixel), no traversal-stack overflow."""
    from strelka_amd import capi

    sc = scenes.hair_standin()
    arr = sc.arrays()
    W, H, SPP = 1920, 1080, 1024
    few, stf = _gpu_frame(arr, sc.getCamera(), W, H, 4, 3, total=SPP)
    ctx = capi.Context(0)
    ctx.set_scene(arr)
    ctx.resize(W, H)
    ctx.render_subframes(S.frame_params(sc.getCamera(), W, H, subframe_index=0, samples_this_launch=1, spp_total=SPP, max_depth=3), SPP, None)
    st = ctx.stats()
    full, dif, spec = ctx.read_accum(), ctx.read_aov(0), ctx.read_aov(1)
    ctx.close()
    assert st["stack_overflows"] == 0
    for img in (full, dif, spec):
        assert np.isfinite(img).all() and img[..., :3].min() >= 0.0
    assert W * H * SPP <= st["rays_radiance"] <= W * H * SPP * 3 and st["rays_shadow"] <= st["rays_radiance"]
    assert abs(st["rays_radiance"] / SPP - stf["rays_radiance"] / 4) < 0.01 * stf["rays_radiance"] / 4
    assert abs(full[..., :3].mean() - few[..., :3].mean()) < 0.1 * few[..., :3].mean()
    # converged: the 1024-spp image is far smoother than the 4-spp one (mean absolute Laplacian over the hair region)
    def rough(img):
        g = img[..., 1]
        return np.abs(4 * g[1:-1, 1:-1] - g[:-2, 1:-1] - g[2:, 1:-1] - g[1:-1, :-2] - g[1:-1, 2:]).mean()
    assert rough(full) < 0.5 * rough(few)
