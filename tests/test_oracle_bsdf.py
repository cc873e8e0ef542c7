"""The two BSDFs added in round 2, pinned on the CPU oracle by what their published models guarantee (parity with the reference
is unpinned: the arithmetic lives in the closed MDL SDK):
  * df::chiang_hair_bsdf (Chiang et al. 2016 in the pbrt-v3 formulation): white furnace without absorption, energy loss ordered
    by the absorption coefficient, sample/evaluate consistency, pdf integrates to 1 over the sphere;
  * rough dielectric (Walter et al. 2007, GGX, visible-normal sampling) for OmniGlass' frosting_roughness: pdf integrates to 1
    over both hemispheres, sample/evaluate consistency, the smooth limit reproduces Snell's law."""
import ctypes as C
import math

import numpy as np
import pytest

from strelka_amd import scene as S


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def f32(*v):
    return np.array(v, np.float32)


def hair_mat(sigma=(0.0, 0.0, 0.0), r=0.3, rn=0.3, alpha=0.0, dw=0.0, tint=(0.5, 0.5, 0.5), ior=1.55):
    m = np.zeros((), S.MATERIAL)
    m["type"], m["base_color"], m["roughness"], m["ior"] = 3, tint, r, ior
    m["reserved"] = (sigma[0], sigma[1], sigma[2], rn, alpha, dw)
    return m


def glass_mat(rough, color=(1, 1, 1), ior=1.5):
    m = np.zeros((), S.MATERIAL)
    m["type"], m["base_color"], m["roughness"], m["ior"] = 2, color, rough, ior
    return m


def sample(ork, mat, n, k1, xi, inside=0):
    out = np.zeros(8, np.float32)
    ork.ork_bsdf_sample(p(mat), p(n), p(n), p(k1), p(xi), inside, p(out))
    return out


def evaluate(ork, mat, n, k1, k2, inside=0):
    out = np.zeros(7, np.float32)
    ork.ork_bsdf_evaluate_side(p(mat), p(n), p(n), p(k1), p(np.ascontiguousarray(k2, np.float32)), inside, p(out))
    return out


def sphere_integral(fn, nz=360, nphi=720):
    """midpoint rule over the sphere in (cos theta, phi): deterministic, fine enough for the peaky lobes tested here"""
    z = (np.arange(nz) + 0.5) / nz * 2 - 1
    ph = (np.arange(nphi) + 0.5) / nphi * 2 * math.pi
    zz, pp = np.meshgrid(z, ph, indexing="ij")
    r = np.sqrt(1 - zz * zz)
    dirs = np.stack([r * np.cos(pp), r * np.sin(pp), zz], -1).reshape(-1, 3).astype(np.float32)
    return sum(float(fn(d)) for d in dirs) * (4 * math.pi / len(dirs))


@pytest.mark.parametrize("k1", [(0.1, 0.2, 0.97), (0.7, 0.1, 0.7), (-0.5, -0.3, 0.81)])
@pytest.mark.parametrize("alpha", [0.0, 0.05])
def test_hair_white_furnace_and_consistency(ork, k1, alpha):
    """No absorption: A_R + A_TT + A_TRT + A_residual = 1, so the fibre reflects everything it receives; the sampled pdf is the
    evaluated pdf and evaluate = pdf * bsdf_over_pdf; with absorption every channel loses energy, more where sigma_a is larger."""
    ork.ork_bsdf_set_tangent(p(f32(1, 0, 0)))  # fibre along x, surface normal z
    n = f32(0, 0, 1)
    k1 = f32(*k1)
    k1 /= np.linalg.norm(k1)
    rs = np.random.RandomState(3)
    for sigma, lo, hi in [((0, 0, 0), 0.97, 1.03), ((0.2, 0.8, 3.0), 0.03, 0.999)]:
        mat = hair_mat(sigma, alpha=alpha)
        acc = np.zeros(3)
        N = 3000
        for _ in range(N):
            xi = rs.rand(4).astype(np.float32)
            s = sample(ork, mat, n, k1, xi)
            assert int(s[7]) == (2 | 8)  # GLOSSY | REFLECTION: a fibre has no inside
            assert abs(np.linalg.norm(s[:3]) - 1) < 1e-5 and s[6] > 0
            acc += s[3:6]
            ev = evaluate(ork, mat, n, k1, s[:3])
            assert abs(ev[6] - s[6]) <= 2e-3 * max(1.0, s[6]), (ev[6], s[6])
            assert np.allclose(ev[3:6], s[6] * s[3:6], rtol=5e-3, atol=1e-5) and not ev[:3].any()
        mean = acc / N
        assert (mean > lo).all() and (mean < hi).all(), mean
        if sigma[2] > 0:
            assert mean[0] > mean[1] > mean[2]
    # the pdf integrates to 1 over the whole sphere (TT leaves on the far side)
    mat = hair_mat((0.3, 0.3, 0.3), alpha=alpha)
    assert abs(sphere_integral(lambda d: evaluate(ork, mat, n, k1, d)[6], 200, 400) - 1.0) < 0.01


def test_hair_lobes_sit_where_the_model_puts_them(ork):
    """h = 0 (the reference's constant text_coords, closest_hit.cu:445): R comes straight back in azimuth, TT goes straight
    through; light fibres are TT-dominated (forward), dark fibres R-dominated (backward)."""
    ork.ork_bsdf_set_tangent(p(f32(1, 0, 0)))
    n = f32(0, 0, 1)
    k1 = f32(0.0, 0.0, 1.0)
    back, fwd = f32(0.0, 0.05, 0.999), f32(0.0, 0.05, -0.999)
    light, dark = hair_mat((0.02, 0.02, 0.02), r=0.15, rn=0.15), hair_mat((8.0, 8.0, 8.0), r=0.15, rn=0.15)
    e = {k: (evaluate(ork, m, n, k1, back / np.linalg.norm(back))[3], evaluate(ork, m, n, k1, fwd / np.linalg.norm(fwd))[3])
         for k, m in (("light", light), ("dark", dark))}
    assert e["light"][1] > 3 * e["light"][0]  # transmission wins for a blond fibre
    assert e["dark"][0] > 10 * e["dark"][1]  # nothing gets through a black one
    assert e["dark"][0] == pytest.approx(e["light"][0], rel=0.5)  # the R lobe does not depend on the absorption


def test_hair_diffuse_weight(ork):
    ork.ork_bsdf_set_tangent(p(f32(1, 0, 0)))
    n, k1 = f32(0, 0, 1), f32(0.3, 0.2, 0.93)
    k1 /= np.linalg.norm(k1)
    mat = hair_mat((1, 1, 1), dw=0.4, tint=(0.2, 0.5, 0.9))
    k2 = f32(0.1, -0.4, 0.9)
    k2 /= np.linalg.norm(k2)
    ev, ev0 = evaluate(ork, mat, n, k1, k2), evaluate(ork, hair_mat((1, 1, 1)), n, k1, k2)
    assert np.allclose(ev[:3], np.array([0.2, 0.5, 0.9]) * 0.4 * k2[2] / math.pi, rtol=1e-5)
    assert np.allclose(ev[3:6], 0.6 * ev0[3:6], rtol=1e-5)
    assert ev[6] == pytest.approx(0.4 * k2[2] / math.pi + 0.6 * ev0[6], rel=1e-5)
    rs = np.random.RandomState(1)
    kinds = {int(sample(ork, mat, n, k1, rs.rand(4).astype(np.float32))[7]) for _ in range(200)}
    assert kinds == {1 | 8, 2 | 8}


@pytest.mark.parametrize("rough", [0.2, 0.45, 0.8])
@pytest.mark.parametrize("inside", [0, 1])
def test_rough_glass_pdf_and_consistency(ork, rough, inside):
    n = f32(0, 0, 1)
    k1 = f32(0.3, -0.2, 0.93)
    k1 /= np.linalg.norm(k1)
    mat = glass_mat(rough, color=(0.9, 0.95, 1.0))
    rs = np.random.RandomState(5)
    nt = accepted = 0
    acc = np.zeros(3)
    N = 3000
    for _ in range(N):
        xi = rs.rand(4).astype(np.float32)
        s = sample(ork, mat, n, k1, xi, inside)
        if s[7] == 0:
            continue
        accepted += 1
        ev_t = int(s[7])
        assert ev_t in (2 | 8, 2 | 16)
        assert (s[2] > 0) == (ev_t == (2 | 8))  # reflection stays on k1's side, transmission crosses
        nt += ev_t == (2 | 16)
        acc += s[3:6]
        ev = evaluate(ork, mat, n, k1, s[:3], inside)
        assert abs(ev[6] - s[6]) <= 2e-3 * max(1.0, s[6]), (ev[6], s[6])
        assert np.allclose(ev[3:6], s[6] * s[3:6], rtol=5e-3, atol=1e-6)
        assert (s[3:6] <= 1.0 + 1e-4).all()  # G2 / G1 <= 1
    assert nt > 0.5 * N if not inside else nt > 0.2 * N
    assert (acc / N > (0.7 if rough < 0.5 else 0.45)).all()  # little is lost to masking at moderate roughness
    # the pdf integrates, over both hemispheres, to the probability that a sample is accepted (micro-normals whose reflection or
    # refraction ends on the wrong side of the surface are absorbed: rare at low roughness, a few 10 % at 0.8 from inside)
    fine = rough < 0.3  # alpha = 0.04: a 2-degree lobe needs the finer grid
    total = sphere_integral(lambda d: evaluate(ork, mat, n, k1, d, inside)[6], 720 if fine else 360, 1440 if fine else 720)
    assert total < 1.005 and abs(total - accepted / N) < 0.03, (total, accepted / N)


def test_rough_glass_smooth_limit_is_snell(ork):
    n = f32(0, 0, 1)
    k1 = f32(0.0, 0.6, 0.8)
    s = sample(ork, glass_mat(0.012), n, k1, f32(0.5, 0.5, 0.99, 0))
    assert int(s[7]) == (2 | 16) and abs(math.hypot(s[0], s[1]) - 0.6 / 1.5) < 2e-3
    # below the threshold the delta branch answers (specular events, pdf 0): the round-1 behaviour for clear glass
    d = sample(ork, glass_mat(0.0), n, k1, f32(0.5, 0.5, 0.99, 0))
    assert int(d[7]) == (4 | 16) and d[6] == 0 and abs(math.hypot(d[0], d[1]) - 0.4) < 1e-6


def test_pbr_lobes_against_the_published_microfacet_formulas(ork):
    """A9, the OmniPBR-equivalent material (type 1): its glossy lobe is the GGX microfacet BRDF, its diffuse lobe Lambert under the Fresnel layer.
    The oracle's evaluation -- which the HIP code is compared with call by call (`skh_bsdf_probe`) -- is held here against the PUBLISHED terms, typed
    independently in fp64: D = a^2 / (pi ((n.h)^2 (a^2 - 1) + 1)^2) (Walter et al. 2007), Lambda = (sqrt(1 + a^2 tan^2) - 1) / 2 and the
    height-correlated G2 = 1 / (1 + Lambda_o + Lambda_i) (Heitz 2014), F = f0 + (1 - f0)(1 - o.h)^5 (Schlick 1994), visible-normal pdf
    G1 D / (4 n.o) (Heitz 2018); a = max(roughness, 0.05)^2, f0 = lerp(0.08 specular, base, metallic), diffuse albedo = base (1 - metallic), lobe
    selection 0.5 + 0.5 metallic (include/strelka_hip.h).  400 random (material, normal, k1, k2) cases, both lobes times cos and the pdf: 2e-5."""
    rs = np.random.RandomState(9)

    def unit(v):
        return v / np.linalg.norm(v)

    checked = 0
    for _ in range(400):
        base = rs.uniform(0.05, 0.95, 3)
        rough, metallic, specular = rs.uniform(0.02, 1.0), rs.choice([0.0, 1.0, rs.uniform()]), rs.uniform(0.0, 1.0)
        m = np.zeros((), S.MATERIAL)
        m["type"], m["base_color"], m["roughness"], m["metallic"], m["specular"], m["ior"] = 1, base, rough, metallic, specular, 1.5
        n = unit(rs.normal(size=3))
        # both directions in the upper hemisphere of n
        k1 = unit(rs.normal(size=3))
        k2 = unit(rs.normal(size=3))
        k1 = k1 if k1 @ n > 0 else -k1
        k2 = k2 if k2 @ n > 0 else -k2
        if k1 @ n < 0.05 or k2 @ n < 0.05:
            continue
        n32, k132, k232 = n.astype(np.float32), k1.astype(np.float32), k2.astype(np.float32)
        got = evaluate(ork, m, n32, k132, k232)
        # published terms in fp64, on the float32 inputs the oracle saw
        n_, o, i = n32.astype(np.float64), k132.astype(np.float64), k232.astype(np.float64)
        n_, o, i = unit(n_), unit(o), unit(i)
        base32 = np.asarray(m["base_color"], np.float64)
        rough32, metal32, spec32 = float(m["roughness"]), float(m["metallic"]), float(m["specular"])
        a = max(rough32, 0.05) ** 2
        h = unit(o + i)
        no, ni, nh, oh = n_ @ o, n_ @ i, n_ @ h, max(o @ h, 0.0)
        f0 = 0.08 * spec32 + (base32 - 0.08 * spec32) * metal32
        Fh = f0 + (1.0 - f0) * (1.0 - oh) ** 5
        Fo = f0 + (1.0 - f0) * (1.0 - no) ** 5
        D = a * a / (math.pi * (nh * nh * (a * a - 1.0) + 1.0) ** 2)

        def lam(c):
            return 0.5 * (math.sqrt(1.0 + a * a * (1.0 - c * c) / (c * c)) - 1.0)

        G2, G1 = 1.0 / (1.0 + lam(no) + lam(ni)), 1.0 / (1.0 + lam(no))
        glossy = Fh * D * G2 / (4.0 * no)  # f cos(i): the n.i of the BRDF's denominator cancels
        diffuse = base32 * (1.0 - metal32) * (1.0 - Fo) * ni / math.pi
        ps = 0.5 + 0.5 * metal32
        pdf = ps * G1 * D / (4.0 * no) + (1.0 - ps) * ni / math.pi
        assert np.allclose(got[0:3], diffuse, rtol=2e-5, atol=1e-7), (got[0:3], diffuse)
        assert np.allclose(got[3:6], glossy, rtol=2e-4, atol=1e-6), (got[3:6], glossy, a, nh)  # (D's denominator cancels near the peak at small alpha)
        assert abs(got[6] - pdf) <= 2e-4 * pdf + 1e-6, (got[6], pdf)
        checked += 1
    assert checked > 300


@pytest.mark.parametrize("inside", [0, 1])
def test_smooth_glass_reflects_with_the_exact_fresnel_probability(ork, inside):
    """OmniGlass without frosting: the reflection event is chosen with probability F, the unpolarised Fresnel reflectance of a dielectric
    interface -- F = ((n1 c1 - n2 c2) / (n1 c1 + n2 c2))^2 / 2 + ((n1 c2 - n2 c1) / (n1 c2 + n2 c1))^2 / 2 with Snell's c2, 1 beyond the critical
    angle (Born & Wolf) -- located here by bisection on the selection variable xi.z and compared with that formula in fp64, from outside and
    from inside (total internal reflection included); the refracted direction obeys Snell's law to 1e-6."""
    ior = 1.5
    mat = glass_mat(0.0, ior=ior)
    n = f32(0, 0, 1)
    n1, n2 = (ior, 1.0) if inside else (1.0, ior)
    for c1 in (1.0, 0.9, 0.7, 0.5, 0.3, 0.15, 0.05):
        k1 = f32(math.sqrt(1 - c1 * c1), 0, c1)
        s2 = n1 / n2 * math.sqrt(1 - c1 * c1)
        if s2 >= 1.0:
            want = 1.0
        else:
            c2 = math.sqrt(1 - s2 * s2)
            rs_ = ((n1 * c1 - n2 * c2) / (n1 * c1 + n2 * c2)) ** 2
            rp_ = ((n1 * c2 - n2 * c1) / (n1 * c2 + n2 * c1)) ** 2
            want = 0.5 * (rs_ + rp_)
        lo, hi = 0.0, 1.0  # reflect iff xi.z < F
        for _ in range(30):
            mid = 0.5 * (lo + hi)
            out = sample(ork, mat, n, k1, f32(0.3, 0.6, mid, 0.0), inside)
            reflected = out[2] > 0  # k2.z on the side of k1
            if reflected:
                lo = mid
            else:
                hi = mid
        got = 0.5 * (lo + hi)
        assert abs(got - want) < 2e-6, (inside, c1, got, want)
        if want < 1.0:
            out = sample(ork, mat, n, k1, f32(0.3, 0.6, 0.999999, 0.0), inside)
            assert out[2] < 0 and abs(math.hypot(out[0], out[1]) - s2) < 1e-6  # Snell: sin(t) = n1 / n2 sin(i)
