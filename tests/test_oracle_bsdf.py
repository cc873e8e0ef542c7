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


# ----------------------------------------------------------------------------------------------------------------------
# Formula-level second opinions (round 6): the two BSDFs above were held by properties only.  Below they are typed again,
# INDEPENDENTLY of oracle/ork_bsdf.h and strelka_amd/csrc/skh_device.h, in float64 from the publications -- other
# parametrisations where the literature offers one (Walter's g/c Fresnel form instead of rs/rp, tan-form Smith terms,
# absolute indices instead of their ratio, sin(2^k alpha) from libm instead of the doubling recurrence, factorial series /
# scipy's I0 instead of Horner constants) -- and compared call by call on random (k1, k2, parameters).
# ----------------------------------------------------------------------------------------------------------------------
def _unit(v):
    return v / np.linalg.norm(v)


def _fresnel_walter(c, eta_i, eta_t):
    """Walter et al. 2007, eq. 22: F = 1/2 (g-c)^2/(g+c)^2 (1 + (c(g+c)-1)^2 / (c(g-c)+1)^2), g^2 = eta_t^2/eta_i^2 - 1 + c^2;
    g imaginary = total internal reflection."""
    g2 = (eta_t / eta_i) ** 2 - 1.0 + c * c
    if g2 <= 0.0:
        return 1.0
    g = math.sqrt(g2)
    return 0.5 * ((g - c) / (g + c)) ** 2 * (1.0 + ((c * (g + c) - 1.0) / (c * (g - c) + 1.0)) ** 2)


def _walter_rough_dielectric(alpha, n_i, n_o, tint, wo, wi):
    """f * |cos(wi)| and the sampling pdf of a rough dielectric interface, local frame z = macro normal on wo's side.
    Walter et al. 2007: eq. 20 (reflection) f_r = F D G / (4 |i.n| |o.n|); eq. 21 (refraction)
    f_t = |i.h| |o.h| / (|i.n| |o.n|) * eta_o^2 (1 - F) G D / (eta_i (i.h) + eta_o (o.h))^2 with h = -(eta_i i + eta_o o) normalised (eq. 16);
    eq. 33 D_GGX = alpha^2 / (pi cos^4(t_m) (alpha^2 + tan^2(t_m))^2); half-vector Jacobians eq. 14 (1 / (4 |o.h|)) and eq. 17
    (eta_o^2 |o.h| / (eta_i (i.h) + eta_o (o.h))^2).  Masking-shadowing: Smith with Lambda = (-1 + sqrt(1 + alpha^2 tan^2)) / 2 (Walter eq. 34
    is G1 = 2 / (1 + sqrt(1 + alpha^2 tan^2)) = 1 / (1 + Lambda)), height-correlated G = 1 / (1 + Lambda_i + Lambda_o) (Heitz 2014, eq. 99).
    Sampling: visible normals D_wo(h) = G1(wo) max(0, wo.h) D(h) / wo.z (Heitz 2018, eq. 1), reflection chosen with probability F.
    i := wo (the known direction, medium n_i), o := wi (medium n_i for reflection, n_o for refraction).  No radiance scaling
    (n_o^2 / n_i^2) -- include/strelka_hip.h states that for the GLASS type."""
    ci, co = wo[2], wi[2]
    if ci <= 0.0 or co == 0.0:
        return np.zeros(3), 0.0
    reflect = co > 0.0

    def tan2(c):
        return (1.0 - c * c) / (c * c)

    def lam(c):
        return 0.5 * (-1.0 + math.sqrt(1.0 + alpha * alpha * tan2(c)))

    if reflect:
        h = _unit(wo + wi)
    else:
        h = -(n_i * wo + n_o * wi)
        if not np.linalg.norm(h) > 0:
            return np.zeros(3), 0.0
        h = _unit(h)
    if h[2] < 0:
        h = -h
    ih, oh = float(wo @ h), float(wi @ h)
    if ih <= 0.0 or (oh <= 0.0 if reflect else oh >= 0.0):
        return np.zeros(3), 0.0
    F = _fresnel_walter(min(ih, 1.0), n_i, n_o)
    ct2 = 1.0 - (n_i / n_o) ** 2 * (1.0 - min(ih, 1.0) ** 2)  # cos^2 of the refracted angle: -> 0 at the critical angle, where dF / dc diverges
    _walter_rough_dielectric.kappa_f = (n_i / n_o) ** 2 / ct2 if ct2 > 0 else 0.0
    cm = h[2]
    D = alpha ** 2 / (math.pi * cm ** 4 * (alpha ** 2 + tan2(cm)) ** 2)
    G = 1.0 / (1.0 + lam(ci) + lam(abs(co)))
    G1 = 1.0 / (1.0 + lam(ci))
    Dvis = G1 * ih * D / ci
    _walter_rough_dielectric.kappa = 1.0
    if reflect:
        f = F * D * G / (4.0 * ci * co)
        return np.full(3, f * co), F * Dvis / (4.0 * oh)
    den = (n_i * ih + n_o * oh) ** 2
    f = ih * abs(oh) / (ci * abs(co)) * n_o ** 2 * (1.0 - F) * G * D / den
    _walter_rough_dielectric.kappa = (n_i * ih + n_o * abs(oh)) / abs(n_i * ih + n_o * oh)  # cancellation in eq. 21's denominator (and in |h|)
    return np.asarray(tint, np.float64) * f * abs(co), (1.0 - F) * Dvis * n_o ** 2 * abs(oh) / den


def test_rough_glass_against_walter_2007_in_fp64(ork):
    """>= 500 random (normal, k1, k2, roughness, ior, colour) cases per side of the interface: evaluate() and the weight / pdf that sample()
    returns for the direction it chose, against the float64 typing above at 1e-4 relative (the oracle is fp32; atol covers lobe tails).  Near the
    peak of a narrow lobe D's denominator (n.h)^2 (alpha^2 - 1) + 1 cancels down to alpha^2, so one fp32 rounding of (n.h)^2 is worth 6e-8 / alpha^2
    relative, squared: the bar is 1e-4 + 4e-7 / alpha^2 (1e-4 ... 1.5e-4 for roughness >= 0.3; 2.4e-3 at roughness 0.115), plus, for refraction,
    1e-6 x the cancellation in eq. 21's denominator eta_i (i.h) + eta_o (o.h) (opposite signs; the same cancellation shortens h before it is
    normalised) -- kappa = (|eta_i i.h| + |eta_o o.h|) / |sum|, typically 1 ... 20, and 3e-7 x eta^2 / cos^2(theta_t)
    for the Fresnel term next to the critical angle (cos(theta_t) is the root of a difference that cancels there; 1 - F is proportional to it).
    1 500 cases: all but a handful sit below 1e-4."""
    rs = np.random.RandomState(20)
    checked = {0: 0, 1: 0}
    sampled = 0
    for it in range(1500):
        inside = it & 1
        rough = float(np.float32(rs.uniform(0.08, 1.0)))
        ior = float(np.float32(rs.uniform(1.1, 2.2)))
        tint = rs.uniform(0.2, 1.0, 3).astype(np.float32)
        mat = glass_mat(rough, color=tuple(tint), ior=ior)
        n = _unit(rs.normal(size=3)).astype(np.float32)
        k1 = _unit(rs.normal(size=3))
        k1 = (k1 if k1 @ n > 0 else -k1).astype(np.float32)
        if k1 @ n < 0.05:
            continue
        n64 = _unit(n.astype(np.float64))
        alpha = max(float(np.float32(rough) * np.float32(rough)), 1e-4)
        n_i, n_o = (ior, 1.0) if inside else (1.0, ior)
        tol0 = 1e-4 + 4e-7 / (alpha * alpha)
        # a frame around n: the model is isotropic, any tangent pair serves
        t1 = _unit(np.cross(n64, [0.0, 0.0, 1.0] if abs(n64[2]) < 0.9 else [1.0, 0.0, 0.0]))
        t2 = np.cross(n64, t1)

        def local(v):
            v = _unit(v.astype(np.float64))
            return np.array([v @ t1, v @ t2, v @ n64])

        # (1) evaluate() on a random k2 -- half of them on the far side
        k2 = _unit(rs.normal(size=3)).astype(np.float32)
        if abs(k2 @ n) > 0.05:
            got = evaluate(ork, mat, n, k1, k2, inside)
            f_cos, pdf = _walter_rough_dielectric(alpha, n_i, n_o, tint, local(k1), local(k2))
            tol = tol0 + 1e-6 * _walter_rough_dielectric.kappa + 3e-7 * _walter_rough_dielectric.kappa_f
            assert not got[:3].any()
            assert np.allclose(got[3:6], f_cos, rtol=tol, atol=1e-7), (got[3:6], f_cos, rough, ior, inside)
            assert abs(got[6] - pdf) <= tol * pdf + 1e-7, (got[6], pdf)
            checked[inside] += pdf > 0
        # (2) what sample() hands back for the direction it chose
        s = sample(ork, mat, n, k1, rs.rand(4).astype(np.float32), inside)
        if s[7] == 0:
            continue
        f_cos, pdf = _walter_rough_dielectric(alpha, n_i, n_o, tint, local(k1), local(s[:3]))
        tol = tol0 + 1e-6 * _walter_rough_dielectric.kappa + 3e-7 * _walter_rough_dielectric.kappa_f
        assert abs(s[6] - pdf) <= tol * pdf + 1e-7, (s[6], pdf, rough, ior, inside)
        assert np.allclose(s[3:6] * s[6], f_cos, rtol=tol, atol=1e-7)
        assert (int(s[7]) == (2 | 16)) == (local(s[:3])[2] < 0)
        sampled += 1
    assert min(checked.values()) >= 250 and sum(checked.values()) >= 500 and sampled >= 500, (checked, sampled)


def _pbrt_hair(wo, wi, eta, sigma_a, beta_m, beta_n, alpha_rad, h, beta_tt=0.0, beta_trt=0.0, lum=(0.299, 0.587, 0.114), bessel="pbrt"):
    """df::chiang_hair_bsdf in the pbrt-v3 formulation (Pharr, "The implementation of a hair scattering model", 2016; hair.cpp), float64.
    wo, wi in the fibre frame (x along the fibre).  Returns (f * |cos|  [the p-sum: pbrt divides it by |cos(wi)| for its integrator], pdf,
    ap, ap_pdf).  Longitudinal M_p (eq. 7 of Chiang et al.: exp(-sin_i sin_o / v) I0(cos_i cos_o / v) / (2 v sinh(1/v))), azimuthal N_p = logistic
    of scale s trimmed to [-pi, pi] around Phi(p) = 2 p gamma_t - 2 gamma_o + p pi, attenuations A_0 = F, A_1 = (1-F)^2 T, A_2 = A_1 T F,
    A_rest = A_2 F T / (1 - T F); cuticle tilt 2 alpha (R), -alpha (TT), -4 alpha (TRT); v = (0.726 b + 0.812 b^2 + 3.7 b^20)^2, v_TT = v/4,
    v_TRT = 4 v (unless given); s = sqrt(pi/8) (0.265 b + 1.194 b^2 + 5.372 b^22).
    bessel = "pbrt": I0 and log I0 as hair.h approximates them -- the ten-term series sum_i x^(2i) / (4^i (i!)^2), and for x > 12
    x + (-log(2 pi) + log(1/x) + 1/(8x)) / 2; M_p for v <= 0.1 as exp(logI0(a) - b - 1/v + 0.6931 + log(1/(2v))).  These are up to 1.5 % away from the
    true Bessel function (the series at x = 12; the asymptote's 1/(16x) instead of 1/(8x)); bessel = "exact" uses scipy's I0 in M_p's definition."""
    from scipy.special import ive  # exponentially scaled I0: ive(0, a) = I0(a) exp(-a)

    def series(x):
        return sum(x ** (2 * i) / (4.0 ** i * math.factorial(i) ** 2) for i in range(10))

    sin_o, sin_i = min(max(wo[0], -1.0), 1.0), min(max(wi[0], -1.0), 1.0)
    cos_o, cos_i = math.sqrt(max(0.0, 1 - sin_o ** 2)), math.sqrt(max(0.0, 1 - sin_i ** 2))
    phi = math.atan2(wi[2], wi[1]) - math.atan2(wo[2], wo[1])
    sin_t = sin_o / eta
    cos_t = math.sqrt(max(0.0, 1 - sin_t ** 2))
    etap = math.sqrt(eta * eta - sin_o ** 2) / max(cos_o, 1e-6)
    sin_gt = h / etap
    cos_gt = math.sqrt(max(0.0, 1 - sin_gt ** 2))
    gamma_t, gamma_o = math.asin(min(max(sin_gt, -1.0), 1.0)), math.asin(min(max(h, -1.0), 1.0))
    T = np.exp(-np.asarray(sigma_a, np.float64) * (2.0 * cos_gt / max(cos_t, 1e-6)))
    # attenuation
    c = cos_o * math.sqrt(max(0.0, 1 - h * h))
    F = _fresnel_walter(min(max(c, 0.0), 1.0), 1.0, eta)
    ap = [np.full(3, F), (1 - F) ** 2 * T]
    ap.append(ap[1] * T * F)
    ap.append(ap[2] * F * T / (1.0 - T * F))
    y = np.array([np.dot(lum, a) for a in ap])
    ap_pdf = y / y.sum()

    def var(b):
        b = max(b, 0.02)
        return (0.726 * b + 0.812 * b ** 2 + 3.7 * b ** 20) ** 2

    v = [var(beta_m)]
    v.append(var(beta_tt) if beta_tt > 0 else v[0] / 4.0)
    v.append(var(beta_trt) if beta_trt > 0 else 4.0 * v[0])
    v.append(v[2])
    bn = max(beta_n, 0.02)
    s = math.sqrt(math.pi / 8.0) * (0.265 * bn + 1.194 * bn ** 2 + 5.372 * bn ** 22)

    def log_Mp(ci, co, si, so, vv):
        a, b = ci * co / vv, si * so / vv
        if bessel == "pbrt":
            if vv <= 0.1:
                log_i0 = a + 0.5 * (-math.log(2 * math.pi) + math.log(1.0 / a) + 1.0 / (8.0 * a)) if a > 12 else math.log(series(a))
                return log_i0 - b - 1.0 / vv + 0.6931 + math.log(1.0 / (2.0 * vv))
            return math.log(math.exp(-b) * series(a) / (math.sinh(1.0 / vv) * 2.0 * vv))
        # log[ exp(-b) I0(a) / (2 v sinh(1/v)) ], with sinh(x) = exp(x) (1 - exp(-2x)) / 2
        return math.log(ive(0, a)) + a - b - math.log(2.0 * vv) - (1.0 / vv + math.log((1.0 - math.exp(-2.0 / vv)) / 2.0))

    def logistic(x):
        e = math.exp(-abs(x) / s)
        return e / (s * (1 + e) ** 2)

    def cdf(x):
        return 1.0 / (1.0 + math.exp(-x / s))

    def Np(p):
        d = phi - (2 * p * gamma_t - 2 * gamma_o + p * math.pi)
        d = (d + math.pi) % (2 * math.pi) - math.pi
        return logistic(d) / (cdf(math.pi) - cdf(-math.pi))

    tilt = [2.0 * alpha_rad, -alpha_rad, -4.0 * alpha_rad]  # R: theta_o - 2 alpha; TT: + alpha; TRT: + 4 alpha
    f, pdf = np.zeros(3), 0.0
    for p in range(3):
        th = math.asin(sin_o) - tilt[p]
        so, co = math.sin(th), abs(math.cos(th))
        mn = math.exp(log_Mp(cos_i, co, sin_i, so, v[p])) * Np(p)
        f = f + ap[p] * mn
        pdf += ap_pdf[p] * mn
    mr = math.exp(log_Mp(cos_i, cos_o, sin_i, sin_o, v[3])) / (2 * math.pi)
    return f + ap[3] * mr, pdf + ap_pdf[3] * mr, ap, ap_pdf


def _hair_case(rs, low_rough):
    sigma = tuple(np.float32(rs.uniform(0.0, 3.0, 3)) * (rs.rand() < 0.8))
    r = float(np.float32(rs.uniform(*low_rough)))
    rn = float(np.float32(rs.uniform(0.15, 1.0)))
    alpha = float(np.float32(rs.uniform(0.0, 0.09)))  # up to ~5 degrees (pbrt's default is 2)
    ior = float(np.float32(rs.uniform(1.3, 1.8)))
    X = _unit(rs.normal(size=3)).astype(np.float32)
    n = rs.normal(size=3)
    n = _unit(n - (n @ X) * X + 0.2 * rs.normal() * X).astype(np.float32)  # not exactly orthogonal: the frame must orthogonalise it
    return sigma, r, rn, alpha, ior, X, n


def _hair_frame64(X, n):
    X64 = _unit(X.astype(np.float64))
    Z = _unit(n.astype(np.float64) - (n.astype(np.float64) @ X64) * X64)
    return X64, np.cross(Z, X64), Z


def test_hair_against_the_pbrt_v3_formulation_in_fp64(ork):
    """>= 500 random (tangent, normal, k1, k2, roughnesses, cuticle angle, absorption, ior): evaluate()'s glossy value (f |cos|) and pdf against the
    float64 typing above.  Moderate-to-high longitudinal roughness at 1e-4 relative (+ a tail floor); the fp32 evaluation of
    exp(logI0(a) - b - 1/v + ...) cancels ~1/v-sized terms, so fibres down to roughness 0.1 (1/v = 150; 600 for TT) are held at 1e-3.
    Every case is ALSO held against the model with the true Bessel function (scipy) at 2 %: what pbrt's I0 approximations are worth."""
    n_done = 0
    for band, tol, count in (((0.3, 1.0), 1e-4, 450), ((0.1, 0.3), 1e-3, 250)):
        rs = np.random.RandomState(31 + int(band[0] * 10))
        for _ in range(count):
            sigma, r, rn, alpha, ior, X, n = _hair_case(rs, band)
            mat = hair_mat(sigma, r=r, rn=rn, alpha=alpha, ior=ior)
            ork.ork_bsdf_set_tangent(p(X))
            k1 = _unit(rs.normal(size=3)).astype(np.float32)
            k2 = _unit(rs.normal(size=3)).astype(np.float32)
            Xf, Yf, Zf = _hair_frame64(X, n)
            loc = lambda v: np.array([v.astype(np.float64) @ Xf, v.astype(np.float64) @ Yf, v.astype(np.float64) @ Zf])
            if abs(loc(k1)[0]) > 0.98 or abs(loc(k2)[0]) > 0.98:
                continue  # grazing along the fibre: cos_theta -> 0 and the 1e-6 clamps take over
            got = evaluate(ork, mat, n, k1, k2)
            f, pdf, _, _ = _pbrt_hair(loc(k1), loc(k2), ior, sigma, r, rn, alpha, 0.0)
            assert not got[:3].any()
            assert np.allclose(got[3:6], f, rtol=tol, atol=2e-6 + 1e-6 * f.max()), (got[3:6], f, r, rn, alpha, sigma)
            assert abs(got[6] - pdf) <= tol * pdf + 2e-6, (got[6], pdf, r, rn)
            fx, pdfx, _, _ = _pbrt_hair(loc(k1), loc(k2), ior, sigma, r, rn, alpha, 0.0, bessel="exact")
            assert np.allclose(got[3:6], fx, rtol=2e-2, atol=2e-6) and abs(got[6] - pdfx) <= 2e-2 * pdfx + 2e-6, (got[3:], fx, pdfx)
            n_done += 1
    assert n_done >= 600


def test_hair_terms_one_by_one_against_fp64(ork):
    """The pieces a consistent-but-wrong implementation could hide, each made visible through the public probe:
    * A_p for p = 0..2 + residual: with an isotropic azimuth (rn -> large s) and sigma chosen per channel, the lobe ENERGIES follow from the
      sampled weights' mean per selected lobe -- here simpler: evaluate at a direction where one lobe dominates and compare with that lobe alone;
    * the cuticle tilt: with alpha != 0 the R peak (in theta_i, at phi = 0 for h = 0) sits at theta_i = -theta_o + 2 alpha, TT (phi = pi) at
      -theta_o - alpha: located by a scan of evaluate() and compared with the angles the model states;
    * the diffuse_reflection_weight mix is covered by test_hair_diffuse_weight."""
    Xt, n = f32(1, 0, 0), f32(0, 0, 1)
    ork.ork_bsdf_set_tangent(p(Xt))
    theta_o = 0.3
    k1 = f32(math.sin(theta_o), 0.0, math.cos(theta_o))  # phi_o = atan2(z, y) = pi/2
    for alpha in (0.0, 0.035, 0.08):
        # dark fibre: R only; scan theta_i in the plane phi_i = phi_o (dphi = 0 = Phi(0) for h = 0)
        dark = hair_mat((30.0, 30.0, 30.0), r=0.12, rn=0.2, alpha=alpha)
        th = np.linspace(-1.2, 0.6, 3601)
        vals = [evaluate(ork, dark, n, k1, f32(math.sin(t), 0.0, math.cos(t)))[3] for t in th]
        peak_r = th[int(np.argmax(vals))]
        # M_p(v) peaks where cos(theta_i + theta_o') is maximal, i.e. at theta_i = -theta_o' (small v), theta_o' = theta_o - 2 alpha
        assert abs(peak_r - (-(theta_o - 2 * alpha))) < 4e-3, (alpha, peak_r)
        # clear fibre, forward side (phi_i = phi_o + pi): TT dominates; theta_o' = theta_o + alpha
        clear = hair_mat((0.0, 0.0, 0.0), r=0.12, rn=0.2, alpha=alpha)
        vals = [evaluate(ork, clear, n, k1, f32(math.sin(t), 0.0, -math.cos(t)))[3] for t in th]
        peak_tt = th[int(np.argmax(vals))]
        assert abs(peak_tt - (-(theta_o + alpha))) < 6e-3, (alpha, peak_tt)
    # attenuations: per-channel absorption -> per-channel A_p; compare the whole p-sum AND its R-only / TT-dominated limits with fp64
    sigma = (0.1, 0.7, 2.5)
    mat = hair_mat(sigma, r=0.4, rn=0.4, alpha=0.03, ior=1.55)
    loc = lambda v: np.array([float(v[0]), float(v[1]), float(v[2])])
    for k2 in (f32(0.1, 0.2, 0.97), f32(-0.2, 0.1, -0.97), f32(0.3, 0.9, 0.1), f32(-0.5, -0.8, 0.2)):
        k2 = (k2 / np.linalg.norm(k2)).astype(np.float32)
        got = evaluate(ork, mat, n, k1, k2)
        f, pdf, ap, ap_pdf = _pbrt_hair(loc(k1), loc(k2), 1.55, sigma, 0.4, 0.4, 0.03, 0.0)
        assert np.allclose(got[3:6], f, rtol=1e-4) and abs(got[6] - pdf) < 1e-4 * pdf
    # energy split: A_0 + A_1 + A_2 + A_rest = F + (1-F)^2 T / (1 - T F) (geometric series), = 1 when T = 1
    _, _, ap, ap_pdf = _pbrt_hair(loc(k1), loc(k1), 1.55, (0, 0, 0), 0.4, 0.4, 0.0, 0.0)
    assert np.allclose(sum(ap), 1.0, atol=1e-12) and abs(sum(ap_pdf) - 1) < 1e-12


def test_hair_sampler_against_the_pbrt_v3_formulation_in_fp64(ork):
    """sample(): the lobe choice (cumulative A_p luminances against xi.z), the longitudinal draw cos(theta) = 1 + v log(u + (1 - u) exp(-2 / v)),
    sin(theta_i) = -cos(theta) sin(theta_o') + sin(theta) cos(2 pi u1) cos(theta_o'), the azimuth Phi(p) + trimmed-logistic inverse CDF
    (-s log(1 / (u k + cdf(-pi)) - 1)), residual lobe uniform; the returned pdf and weight (f |cos| / pdf) -- typed in float64 from hair.cpp's
    Sample_f.  Direction within 2e-4 (5e-3 near lobe-selection / clamp boundaries are skipped), pdf and weight at 1e-3."""
    rs = np.random.RandomState(77)
    done = 0
    for _ in range(900):
        sigma, r, rn, alpha, ior, X, n = _hair_case(rs, (0.25, 1.0))
        mat = hair_mat(sigma, r=r, rn=rn, alpha=alpha, ior=ior)
        ork.ork_bsdf_set_tangent(p(X))
        k1 = _unit(rs.normal(size=3)).astype(np.float32)
        Xf, Yf, Zf = _hair_frame64(X, n)
        wo = np.array([k1.astype(np.float64) @ Xf, k1.astype(np.float64) @ Yf, k1.astype(np.float64) @ Zf])
        if abs(wo[0]) > 0.98:
            continue
        xi = rs.rand(4).astype(np.float32)
        xi[0] = max(xi[0], 1e-3)
        s_ = sample(ork, mat, n, k1, xi)
        if s_[7] == 0:
            continue
        _, _, ap, ap_pdf = _pbrt_hair(wo, wo, ior, sigma, r, rn, alpha, 0.0)
        cum = np.cumsum(ap_pdf)
        u = float(xi[2])
        if np.abs(cum[:3] - u).min() < 1e-4:
            continue  # lobe boundary: fp32 and fp64 may choose differently
        pl = int(np.searchsorted(cum[:3], u, side="right"))
        b = max(r, 0.02)
        v0 = (0.726 * b + 0.812 * b ** 2 + 3.7 * b ** 20) ** 2
        v = [v0, v0 / 4, 4 * v0, 4 * v0][pl]
        tilt = [2.0 * alpha, -alpha, -4.0 * alpha, 0.0][pl]
        th = math.asin(min(max(wo[0], -1), 1)) - tilt
        so, co = math.sin(th), abs(math.cos(th))
        u0 = max(float(xi[0]), 1e-5)
        ct = 1.0 + v * math.log(u0 + (1.0 - u0) * math.exp(-2.0 / v))
        st = math.sqrt(max(0.0, 1 - ct * ct))
        sin_i = -ct * so + st * math.cos(2 * math.pi * float(xi[1])) * co
        cos_i = math.sqrt(max(0.0, 1 - sin_i ** 2))
        bn = max(rn, 0.02)
        sc = math.sqrt(math.pi / 8.0) * (0.265 * bn + 1.194 * bn ** 2 + 5.372 * bn ** 22)
        cdf = lambda x: 1.0 / (1.0 + math.exp(-x / sc))
        u3 = float(xi[3])
        if pl < 3:
            k = cdf(math.pi) - cdf(-math.pi)
            x = -sc * math.log(1.0 / (u3 * k + cdf(-math.pi)) - 1.0)
            dphi = pl * math.pi + min(max(x, -math.pi), math.pi)  # h = 0: gamma_o = gamma_t = 0
        else:
            dphi = 2 * math.pi * u3
        phi_i = math.atan2(wo[2], wo[1]) + dphi
        wi = np.array([sin_i, cos_i * math.cos(phi_i), cos_i * math.sin(phi_i)])
        k2 = wi[0] * Xf + wi[1] * Yf + wi[2] * Zf
        cond = 1.0 / max(st, 1e-3) + 1.0 / max(cos_i, 1e-3)  # the draw's conditioning: d(sin_i) / d(cos theta) and the 1 / cos_i of the azimuth part
        assert np.abs(s_[:3] - k2).max() < 2e-5 * (1 + cond) / max(min(v, 1.0), 0.05), (s_[:3], k2, pl, r, xi)
        f, pdf, _, _ = _pbrt_hair(wo, np.array([s_[:3].astype(np.float64) @ Xf, s_[:3].astype(np.float64) @ Yf, s_[:3].astype(np.float64) @ Zf]),
                                  ior, sigma, r, rn, alpha, 0.0)
        assert abs(s_[6] - pdf) <= 1e-3 * pdf + 2e-6, (s_[6], pdf)
        assert np.allclose(s_[3:6] * s_[6], f, rtol=1e-3, atol=2e-6)
        done += 1
    assert done >= 500, done


def test_hair_bessel_series_is_pbrts_and_close_to_the_true_I0(ork):
    """pbrt-v3's I0 is a ten-term power series (hair.h); M_p uses it directly for v > 0.1 (a <= 10) and through log for a <= 12.  Typed here from
    the series' definition sum_i x^(2i) / (4^i (i!)^2) with math.factorial -- a mistyped constant in the Horner form would show -- by evaluating M_p
    through the probe at sin = 0 (b = 0): M_p = I0(cos_i cos_o / v) / (2 v sinh(1 / v)).  The series itself is within 3.5e-3 of the true I0 at
    a = 10 and 1e-6 below a = 5 (scipy.special.i0): stated, since the check above uses the true I0."""
    from scipy.special import i0

    series = lambda x: sum(x ** (2 * i) / (4.0 ** i * math.factorial(i) ** 2) for i in range(10))
    for a in (0.5, 2.0, 5.0):
        assert abs(series(a) / i0(a) - 1) < 1e-6
    assert abs(series(10.0) / i0(10.0) - 1) < 3.5e-3
    # through the probe: a fibre along x, k1 = k2 = z (theta = 0, phi = 0 -> R lobe centred), dark fibre so that only R matters
    Xt, n = f32(1, 0, 0), f32(0, 0, 1)
    ork.ork_bsdf_set_tangent(p(Xt))
    k = f32(0, 0, 1)
    for r in (0.45, 0.6, 0.9):  # v = 0.24, 0.54, 1.8: the direct branch
        b = r
        v = (0.726 * b + 0.812 * b ** 2 + 3.7 * b ** 20) ** 2
        assert v > 0.1
        got = evaluate(ork, hair_mat((50.0, 50.0, 50.0), r=r, rn=0.3, ior=1.55), n, k, k)[3]
        F = _fresnel_walter(1.0, 1.0, 1.55)
        bn = 0.3
        s = math.sqrt(math.pi / 8.0) * (0.265 * bn + 1.194 * bn ** 2 + 5.372 * bn ** 22)
        n0 = (0.25 / s) / (1.0 / (1.0 + math.exp(-math.pi / s)) - 1.0 / (1.0 + math.exp(math.pi / s)))  # trimmed logistic at 0
        want = F * series(1.0 / v) / (2.0 * v * math.sinh(1.0 / v)) * n0
        assert abs(got / want - 1) < 2e-5, (r, got, want)
