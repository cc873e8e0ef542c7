"""The HIP device code against the REFERENCE's own numbers, directly.

tests/golden/* were written by oracle/_ref/ref_golden -- the reference's RandomSampler.h, Lights.h, postprocessing/Utils.h and
sutil compiled on the host from /root/reference (generator oracle/ref_golden.cpp, recipe oracle/Makefile).  The CPU suite holds the
ORACLE against them (tests/test_oracle_golden.py); here `skh_unit_probe` runs the product's device functions (skh_device.h:
sampler_random / sampler_random_lut / init_sampler / sobol_uint, the four light samplers, get_light_pdf, calc_light_normal / area,
mis_weight_balance, accumulate, tonemap / inverse_tonemap) on the GPU, one call per fixture record.

Bars: bit-exact for everything that is integer work or fp32 + - * / sqrt (sampler, Sobol words, sample index, uniform rectangle
sampling, rect / sphere pdfs, normals, areas, MIS weight, the accumulation sequence, the tonemap pair); the CPU test's ulp bars where
libm transcendentals enter (sphere / distant sampling: sin, cos; spherical rectangle: acos chains; distant pdf: cos).
Reference code matched: RandomSampler.h:130-226, Lights.h:28-362, OptixRender.cu:60-78, postprocessing/Utils.h:5-14.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def g(name, dtype):
    return np.fromfile(os.path.join(G, name), dtype=dtype)


@pytest.fixture(scope="module")
def ctx():
    from strelka_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def _lights():
    raw = g("lights_def.f32", np.float32).reshape(3, 28)
    return [np.ascontiguousarray(r) for r in raw]


def _close_ulp(a, b, ulps):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    tol = ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32))
    return both_nan | (np.abs(a.astype(np.float64) - b.astype(np.float64)) <= tol) | (a == b)


def test_sampler_bit_exact_on_the_device(ctx):
    """random<D> for every (x, y, sample, depth, dim) tuple of the fixture: Morton index, murmur hash, Laine-Karras Owen scramble,
    Sobol word, the % 5 dimension aliasing -- u32 work, bit for bit; and the LDS-table path k_shade really uses gives the same bits."""
    inp = g("sampler_in.u32", np.uint32).reshape(-1, 5)
    want = g("sampler_out.f32", np.float32).view(np.uint32)
    idx = g("sampler_idx.u32", np.uint32)
    out = ctx.unit_probe("sampler", inp, param=64)
    assert np.array_equal(out[:, 0], want)
    assert np.array_equal(out[:, 1], want)  # sampler_random_lut (byte-folded table in LDS)
    assert np.array_equal(out[:, 2], idx)
    # dims 5..9 alias dims 0..4 at every depth
    v = out[:, 0].reshape(-1, 10)
    assert np.array_equal(v[:, :5], v[:, 5:])


def test_sampler_largest_index_does_not_wrap_on_the_device(ctx):
    big = g("sampler_big.u32", np.uint32)
    out = ctx.unit_probe("sampler", np.array([[3839, 2159, 255, 0, 0]], np.uint32), param=256)
    assert int(out[0, 2]) == int(big[0]) == int(big[1]) * 256 + 255


def test_sobol_words_bit_exact_on_the_device(ctx):
    want = g("sobol_uint.u32", np.uint32)
    rec = np.array([[(i * 2654435761 + d) & 0xFFFFFFFF, d] for d in range(5) for i in range(64)], np.uint32)
    out = ctx.unit_probe("sobol", rec)
    assert np.array_equal(out[:, 0], want)


@pytest.mark.parametrize("method,fname,ulps", [(0, "lights_rect_uniform.f32", 0), (1, "lights_rect_sph.f32", -1),
                                               (2, "lights_sphere.f32", 4), (3, "lights_distant.f32", 4)])
def test_light_sampling_on_the_device(ctx, method, fname, ulps):
    rect, sph, dist = _lights()
    light = {0: rect, 1: rect, 2: sph, 3: dist}[method]
    inp = g("lights_in.f32", np.float32).reshape(-1, 5)
    want = g(fname, np.float32).reshape(-1, 12)
    out = ctx.unit_probe("light_sample", inp, param=method, consts=light).view(np.float32)
    if ulps == 0:
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))  # SampleRectLightUniform: + - * / sqrt only
    elif method == 1:
        # the spherical-rectangle sampler chains four acos, then cos / sin through a cancelling sum (Lights.h:97-189): a 1-ulp difference
        # in one acos is amplified ~100x in the sampled point.  Against the fixtures (glibc) the shared skm:: functions measure <= 1.2e-5
        # relative / 160 ulp on a near-zero coordinate; the bar is 3e-5 relative + 1e-6 absolute (was 2e-4 + 2e-5 with the ROCm device
        # library).  Against the CPU restatement the device is exact: test_light_samplers_on_the_device_equal_the_checkers_bit_for_bit.
        assert np.allclose(out, want, rtol=3e-5, atol=1e-6, equal_nan=True)
    else:
        assert _close_ulp(out, want, ulps).all()


def test_light_pdfs_normals_areas_on_the_device(ctx):
    rect, sph, dist = _lights()
    inp = g("lights_in.f32", np.float32).reshape(-1, 5)
    P = np.ascontiguousarray(inp[:, :3])
    want = g("lights_pdf.f32", np.float32).reshape(-1, 4)
    lp = g("lights_rect_uniform.f32", np.float32).reshape(-1, 12)[:, :3]  # the fixture's light hit points (the uniform samples)
    rec = np.ascontiguousarray(np.concatenate([lp, P], axis=1), np.float32)
    pdf = ctx.unit_probe("light_pdf", rec, consts=rect)[:, 0]
    assert np.array_equal(pdf, want[:, 0].copy().view(np.uint32))  # rect: dist^2 / (cos * area), bit-exact
    pdf = ctx.unit_probe("light_pdf", rec, consts=sph)[:, 0].view(np.float32)
    assert np.array_equal(pdf, want[:, 2])  # 1 / 4 pi
    pdf = ctx.unit_probe("light_pdf", rec, consts=dist)[:, 0].view(np.float32)
    assert _close_ulp(pdf, want[:, 3], 2).all()  # cosf(halfAngle)
    nrm = g("lights_normal.f32", np.float32).reshape(-1, 2, 3)
    area = g("lights_area.f32", np.float32)
    for k, l in enumerate((rect, sph)):
        out = ctx.unit_probe("light_normal", P, consts=l).view(np.float32)
        assert np.array_equal(out[:, :3].view(np.uint32), np.ascontiguousarray(nrm[:, k]).view(np.uint32))
        assert np.all(out[:, 3].view(np.uint32) == area[k:k + 1].view(np.uint32))
    out = ctx.unit_probe("light_normal", P[:1], consts=dist).view(np.float32)
    assert out[0, 3].view(np.uint32) == area[2:3].view(np.uint32)[0]


def test_mis_weight_bit_exact_on_the_device(ctx):
    m = g("mis.f32", np.float32).reshape(-1, 3)
    out = ctx.unit_probe("mis", np.ascontiguousarray(m[:, :2]))[:, 0]
    assert np.array_equal(out, m[:, 2].copy().view(np.uint32))


def test_accumulate_sequence_bit_exact_on_the_device(ctx):
    """invTM(lerp(TM(prev), TM(new), 1 / (i + 1))) composed 64 times (OptixRender.cu:60-78): order-dependent, must match bit for bit"""
    vin = g("accum_in.f32", np.float32).reshape(-1, 3)
    want = g("accum_out.f32", np.float32).reshape(-1, 3)
    e = np.full(3, 6.25e-4, np.float32)
    out = ctx.unit_probe("accumulate", vin, param=0, consts=e)
    assert np.array_equal(out, want.view(np.uint32))
    tm = g("tonemap.f32", np.float32).reshape(-1, 2, 3)
    t = ctx.unit_probe("tonemap", vin, consts=e).view(np.float32).reshape(-1, 2, 3)
    assert np.array_equal(t[:, 0].view(np.uint32), np.ascontiguousarray(tm[:, 0]).view(np.uint32))
    back = ctx.unit_probe("tonemap", np.ascontiguousarray(tm[:, 0]), consts=e).view(np.float32).reshape(-1, 2, 3)
    assert np.array_equal(back[:, 1].view(np.uint32), np.ascontiguousarray(tm[:, 1]).view(np.uint32))
    # the SURVEY probe (e = 0.0625)
    probe = g("accum_probe.f32", np.float32)
    e2 = np.full(3, 0.0625, np.float32)
    seq = np.array([[1, 2, 3], [3, 2, 1]], np.float32)
    got = ctx.unit_probe("accumulate", seq, param=0, consts=e2).view(np.float32)[1]
    assert np.array_equal(got.view(np.uint32), probe.view(np.uint32))


def test_unit_probe_refuses_bad_arguments(ctx):
    from strelka_amd import capi

    with pytest.raises(capi.SkhError):
        ctx.unit_probe("light_sample", np.zeros((1, 5), np.float32), param=0, consts=None)  # a light sampler without a light


def test_libm_is_bit_identical_on_the_device(ctx):
    """strelka_amd/csrc/skh_libm.h compiled for gfx950 returns the bits the CPU checker's copy of the same text returns -- sin, cos, acos,
    asin, atan2, exp, log, sinh, pow over 400 k arguments incl. infinities, NaN, zeros, subnormal results.  This is what lets the image
    comparisons of this suite be array_equal: both sides now share every rounding of the render path."""
    from tests import orklib
    from tests.test_libm import cpu_libm, libm_inputs

    rec = libm_inputs()
    want = cpu_libm(rec)
    got = ctx.unit_probe("libm", rec).view(np.float32)
    both_nan = np.isnan(got) & np.isnan(want)
    same = (got.view(np.uint32) == want.view(np.uint32)) | both_nan
    assert same.all(), (int((~same).sum()), rec[np.argwhere(~same)[:5, 0]], np.argwhere(~same)[:5])


def test_libm_on_the_device_against_float64(ctx):
    """An independent guard for the shared transcendentals ON THE DEVICE (VERDICT r5, weak #1a: device and checker compile one text, so a wrong
    polynomial is invisible to every GPU-vs-oracle comparison): what SKH_UNIT_LIBM returns from gfx950 is held directly against numpy float64 at
    the ulp bars of tests/test_libm.py -- sin / cos 1.7, acos 1.3, asin 2.5, exp 1.2, log 1.0, sinh 1.8, atan2 3.5, pow(x, 2.2 | 1/2.2) 4.0 -- with
    no CPU build of skh_libm.h anywhere in the comparison, plus the special values (zeros, infinities, NaN, subnormal results)."""
    from tests.test_libm import ACCURACY_BARS, check_accuracy, check_atan2_and_pow, libm_inputs

    rec = libm_inputs()
    got = ctx.unit_probe("libm", rec).view(np.float32)
    for col, fn, dom, bar in ACCURACY_BARS:
        check_accuracy(rec, got, col, fn, dom, bar)
    check_atan2_and_pow(rec, got)
    f = lambda *v: np.array([v], np.float32)
    o = ctx.unit_probe("libm", np.ascontiguousarray(np.concatenate([f(0.0, 1.0), f(np.nan, 1.0), f(np.inf, 2.0), f(-1.0, 0.0), f(-104.5, 1.0), f(-90.0, 1.0)]))).view(np.float32)
    assert o[0, 0] == 0 and o[0, 1] == 1 and o[0, 3] == 0 and o[0, 5] == 1 and o[0, 6] == -np.inf and o[0, 7] == 0
    assert np.isnan(o[1, :8]).all()
    assert np.isnan(o[2, 0]) and np.isnan(o[2, 1]) and o[2, 5] == np.inf and o[2, 6] == np.inf and o[2, 8] == np.inf
    assert o[3, 2] == np.float32(np.pi) and np.isnan(o[3, 6]) and o[3, 8] == 1.0
    assert o[4, 5] == 0.0 and 0 < o[5, 5] < 1e-38 and abs(o[5, 5] / np.exp(-90.0) - 1) < 1e-5


@pytest.mark.parametrize("method", [0, 1, 2, 3])
def test_light_samplers_on_the_device_equal_the_checkers_bit_for_bit(ctx, method):
    """Beyond the fixtures' ulp bars: device and CPU restatement agree exactly on every light sample (they share skh_libm.h)."""
    import ctypes as C

    from tests import orklib

    ork = orklib.load()
    rect, sph, dist = _lights()
    light = {0: rect, 1: rect, 2: sph, 3: dist}[method]
    inp = g("lights_in.f32", np.float32).reshape(-1, 5)
    P = np.ascontiguousarray(inp[:, :3])
    u = np.ascontiguousarray(inp[:, 3:5])
    want = np.zeros((len(inp), 12), np.float32)
    ork.ork_sample_light(light.ctypes.data_as(C.c_void_p), method, u.ctypes.data_as(C.c_void_p), P.ctypes.data_as(C.c_void_p), len(inp), want.ctypes.data_as(C.c_void_p))
    got = ctx.unit_probe("light_sample", inp, param=method, consts=light).view(np.float32)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), int((~same).sum())
