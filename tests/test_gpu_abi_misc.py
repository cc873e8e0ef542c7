"""Entry points of include/strelka_hip.h that no other GPU test calls directly: ray queries on caller-owned device arrays
(skh_trace_device), the device description bench.py prices its roofs with (skh_get_device_info), the context stream handle
(skh_get_stream) and the device-to-device copy of the accumulator (skh_copy_accum, OptixRender.cpp:1022-1043)."""
import numpy as np
import pytest

from strelka_amd import scene as S, scenes

pytestmark = pytest.mark.gpu


def test_trace_on_device_arrays_equals_trace_on_host_arrays():
    import torch
    from strelka_amd import capi
    sc = scenes.kitchen_standin(seed=3, n_meshes=6, n_instances=30, tri_lo=50, tri_hi=800)
    rays = scenes.random_rays(20000, 5, -4.0, 4.0)
    ctx = capi.Context(0)
    ctx.set_scene(sc.arrays())
    for mode in (0, 1):
        want = ctx.trace(rays, mode)
        d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1).copy()).cuda()
        d_hits = torch.zeros(len(rays) * S.HIT.itemsize, dtype=torch.uint8, device="cuda")
        ctx.trace_device(d_rays.data_ptr(), len(rays), mode, d_hits.data_ptr(), repeat=2)
        ctx.synchronize()
        got = d_hits.cpu().numpy().view(S.HIT)
        assert (got.view(np.uint8) == want.view(np.uint8)).all(), "mode %d" % mode
    assert ctx.stream()  # a hipStream_t the caller may order its own work against
    ctx.close()


def test_device_info_describes_the_gpu_the_roofs_are_priced_on():
    from strelka_amd import capi
    ctx = capi.Context(0)
    info = ctx.device_info()
    ctx.close()
    assert info["wavefront_size"] == 64 and info["simds_per_cu"] == 4
    assert info["compute_units"] >= 1 and info["clock_khz"] > 100000 and info["total_memory_bytes"] > (1 << 30)
    assert "gfx950" in info["name"] or info["name"]  # (the arch string where the runtime reports it, a product name otherwise)


def test_copy_accum_is_the_accumulator():
    import torch
    from strelka_amd import capi
    sc = scenes.cornell_box()
    W = H = 64
    ctx = capi.Context(0)
    ctx.set_scene(sc.arrays())
    ctx.resize(W, H)
    p = S.frame_params(sc.getCamera(), W, H, subframe_index=0, samples_this_launch=1, spp_total=4, max_depth=3)
    ctx.render_subframes(p, 4, None)
    want = ctx.read_accum()
    d = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
    ctx.copy_accum(d.data_ptr())
    ctx.synchronize()
    ctx.close()
    assert (d.cpu().numpy().reshape(H, W, 4) == want.reshape(H, W, 4)).all()


def test_build_info_reports_the_reinsertion_pass():
    """skh_get_build_info: what the last skh_build_accel did to the triangle hierarchy -- rounds, moves, the sum of the internal nodes' box areas
    before and after (the pass must lower it), its time inside ms_build; with the pass off, or on the radix-tree builder, the record says so."""
    from strelka_amd import capi, scenes

    sc = scenes.kitchen_standin(seed=5, n_meshes=12, n_instances=120, tri_lo=100, tri_hi=3000)
    arr = sc.arrays()
    ctx = capi.Context(0)
    ctx.set_scene(arr)
    b = ctx.build_info()
    assert b["triangles"] > 10000 and b["nodes"] > 1000
    assert 1 <= b["reinsert_rounds"] <= 8 and b["reinsert_moves"] > 100 and b["reinsert_min_size"] == 1
    assert 0.5 * b["cost_before"] < b["cost_after"] < 0.99 * b["cost_before"]
    assert 0.0 < b["ms_reinsert"] < b["ms_build"]
    ctx.set_option("reinsert_rounds", 0)
    ctx.set_scene(arr)
    b0 = ctx.build_info()
    assert b0["reinsert_rounds"] == 0 and b0["reinsert_moves"] == 0 and b0["triangles"] == b["triangles"]
    ctx.set_option("reinsert_rounds", 8)
    ctx.set_option("build_quality", 0)  # the Karras radix tree has no per-group parent chain: no reinsertion
    ctx.set_scene(arr)
    assert ctx.build_info()["reinsert_rounds"] == 0
    for name, bad in (("reinsert_rounds", 65), ("reinsert_curve_rounds", -1), ("reinsert_min_size", -3)):
        with pytest.raises(capi.SkhError):
            ctx.set_option(name, bad)
    ctx.close()
