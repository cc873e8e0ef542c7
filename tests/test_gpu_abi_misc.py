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
