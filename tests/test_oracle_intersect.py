"""Known-answer tests for the parts of the path the reference delegates to closed OptiX / MDL SDK:
the oracle's intersectors, BVH and BSDF set are pinned analytically here (parity with the reference is unpinned for
them -- there is no reference arithmetic to compare with)."""
import ctypes as C
import math

import numpy as np
import pytest

from strelka_amd import scene as S
from strelka_amd import scenes
from tests import orklib


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def f32(*v):
    return np.array(v, np.float32)


def test_triangle_known_answers(ork):
    tri = f32(0, 0, 0, 1, 0, 0, 0, 1, 0)
    out = np.zeros(3, np.float32)
    # straight down onto (0.25, 0.5): t = 2, barycentrics (u, v) = weights of p1, p2 (optixGetTriangleBarycentrics)
    assert ork.ork_intersect_triangle(p(f32(0.25, 0.5, 2)), p(f32(0, 0, -1)), 0.0, 1e16, p(tri), p(out)) == 1
    assert np.allclose(out, (2.0, 0.25, 0.5), atol=1e-7)
    # no back-face culling (OPTIX_RAY_FLAG_NONE): same hit from below
    assert ork.ork_intersect_triangle(p(f32(0.25, 0.5, -3)), p(f32(0, 0, 1)), 0.0, 1e16, p(tri), p(out)) == 1
    assert np.allclose(out, (3.0, 0.25, 0.5), atol=1e-7)
    # outside, parallel, behind the origin, beyond tmax, before tmin
    assert ork.ork_intersect_triangle(p(f32(0.8, 0.8, 2)), p(f32(0, 0, -1)), 0.0, 1e16, p(tri), p(out)) == 0
    assert ork.ork_intersect_triangle(p(f32(0.2, 0.2, 2)), p(f32(1, 0, 0)), 0.0, 1e16, p(tri), p(out)) == 0
    assert ork.ork_intersect_triangle(p(f32(0.2, 0.2, 2)), p(f32(0, 0, 1)), 0.0, 1e16, p(tri), p(out)) == 0
    assert ork.ork_intersect_triangle(p(f32(0.2, 0.2, 2)), p(f32(0, 0, -1)), 0.0, 1.5, p(tri), p(out)) == 0
    assert ork.ork_intersect_triangle(p(f32(0.2, 0.2, 2)), p(f32(0, 0, -1)), 2.5, 1e16, p(tri), p(out)) == 0
    # non-normalised direction: t is in units of the direction (instance transforms scale rays, not t)
    assert ork.ork_intersect_triangle(p(f32(0.25, 0.5, 2)), p(f32(0, 0, -4)), 0.0, 1e16, p(tri), p(out)) == 1
    assert np.allclose(out[0], 0.5, atol=1e-7)


def test_watertight_shared_edge(ork):
    """Rays aimed exactly at the shared edge of two triangles hit at least one of them (no cracks)."""
    a = f32(0, 0, 0, 1, 0, 0, 0, 1, 0)
    b = f32(1, 0, 0, 1, 1, 0, 0, 1, 0)
    out = np.zeros(3, np.float32)
    rs = np.random.RandomState(1)
    for _ in range(2000):
        s = np.float32(rs.rand())
        target = np.array([1 - s, s, 0], np.float32)  # a point on the shared edge x + y = 1
        o = (target + rs.normal(size=3) + np.array([0, 0, 3])).astype(np.float32)
        d = (target - o).astype(np.float32)
        ha = ork.ork_intersect_triangle(p(o), p(d), 0.0, 1e16, p(a), p(out))
        hb = ork.ork_intersect_triangle(p(o), p(d), 0.0, 1e16, p(b), p(out))
        assert ha or hb


def straight_curve(r0, r1):
    # cubic B-spline with collinear, equally spaced control points: the segment spans x in [1, 2]
    return f32(0, 0, 0, r0 + (r0 - r1) * 0.0, 1, 0, 0, r0, 2, 0, 0, r1, 3, 0, 0, r1)


def test_curve_known_answers(ork):
    q = f32(0, 0, 0, 0.1, 1, 0, 0, 0.1, 2, 0, 0, 0.1, 3, 0, 0, 0.1)  # constant radius 0.1 cylinder, axis = x
    out = np.zeros(2, np.float32)
    # perpendicular ray through the axis: hits the surface at distance 1 - 0.1, parameter u = x - 1
    assert ork.ork_intersect_curve(p(f32(1.5, 1, 0)), p(f32(0, -1, 0)), 0.0, 1e16, p(q), p(out)) == 1
    assert abs(out[0] - 0.9) < 2e-5 and abs(out[1] - 0.5) < 1e-3
    # offset ray: grazing distance sqrt(r^2 - d^2)
    assert ork.ork_intersect_curve(p(f32(1.25, 1, 0.06)), p(f32(0, -1, 0)), 0.0, 1e16, p(q), p(out)) == 1
    assert abs(out[0] - (1 - math.sqrt(0.1 ** 2 - 0.06 ** 2))) < 5e-5 and abs(out[1] - 0.25) < 2e-3
    # misses: beside the tube, beyond the segment's ends (end caps off), behind, tmax too short
    assert ork.ork_intersect_curve(p(f32(1.5, 1, 0.11)), p(f32(0, -1, 0)), 0.0, 1e16, p(q), p(out)) == 0
    assert ork.ork_intersect_curve(p(f32(2.3, 1, 0)), p(f32(0, -1, 0)), 0.0, 1e16, p(q), p(out)) == 0
    assert ork.ork_intersect_curve(p(f32(0.8, 1, 0)), p(f32(0, -1, 0)), 0.0, 1e16, p(q), p(out)) == 0
    assert ork.ork_intersect_curve(p(f32(1.5, 1, 0)), p(f32(0, 1, 0)), 0.0, 1e16, p(q), p(out)) == 0
    assert ork.ork_intersect_curve(p(f32(1.5, 1, 0)), p(f32(0, -1, 0)), 0.0, 0.8, p(q), p(out)) == 0
    # direction scaling: t halves when the direction doubles
    assert ork.ork_intersect_curve(p(f32(1.5, 1, 0)), p(f32(0, -2, 0)), 0.0, 1e16, p(q), p(out)) == 1
    assert abs(out[0] - 0.45) < 2e-5
    # varying radius (cone-like): radius at u is the B-spline of the control radii
    qv = f32(0, 0, 0, 0.2, 1, 0, 0, 0.2, 2, 0, 0, 0.1, 3, 0, 0, 0.1)
    ev = np.zeros(21, np.float32)
    assert ork.ork_intersect_curve(p(f32(1.5, 1, 0)), p(f32(0, -1, 0)), 0.0, 1e16, p(qv), p(out)) == 1
    ork.ork_curve_eval(p(qv), float(out[1]), p(f32(1.5, 0.15, 0)), p(ev))
    hit = np.array([1.5, 1 - out[0], 0])
    assert abs(np.linalg.norm(hit - ev[:3]) - ev[3]) < 2e-4  # the hit point lies on the swept surface


def test_curve_hits_lie_on_the_surface_for_random_rays(ork):
    rs = np.random.RandomState(2)
    q = f32(0, 0, 0, 0.05, 1, 0.3, 0, 0.06, 2, -0.2, 0.4, 0.04, 3, 0, 0, 0.03)
    out = np.zeros(2, np.float32)
    ev = np.zeros(21, np.float32)
    hits = 0
    for _ in range(3000):
        u = rs.uniform(0.05, 0.95)
        ork.ork_curve_eval(p(q), float(u), p(f32(0, 0, 0)), p(ev))
        target = ev[:3] + rs.normal(size=3) * 0.02
        o = (target + rs.normal(size=3) * 2).astype(np.float32)
        d = (target - o)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        if ork.ork_intersect_curve(p(o), p(d), 0.0, 1e16, p(q), p(out)):
            hits += 1
            ork.ork_curve_eval(p(q), float(out[1]), p(f32(0, 0, 0)), p(ev))
            hp = o + out[0] * d
            assert abs(np.linalg.norm(hp - ev[:3]) - ev[3]) < 3e-4
    assert hits > 1500


def test_a_ray_along_the_tangent_does_not_hit_a_tube_it_passes_far_from(ork):
    """Round 6, `tools/fuzz_render.py` seed 5483: the ray is parallel to the curve's tangent at u = 0.5 (|c'_xy|^2 = 3e-12 in ray-centric coordinates), the tangent
    cone's quadratic degenerates -- b - sqrt(det) cancels to 0 in single precision, det > 0 and |dt| < 5e-5 by rounding --, and the iteration reported a hit at its first
    bisection point, 0.49 away from a curve point of radius 0.096; the GPU's hierarchy never offered that segment to the intersector (its boxes bound the TUBE), the
    checker's did.  A converged point must lie on the tube (`intersect_curve_segment`: radial distance <= 1.0005 x the tangent cone's radius there): the segment is a miss now,
    for this ray and for rays along the tangents of random thick tubes; genuine hits (the tests above) keep their bits."""
    q = f32(-0.9328025, 0.13670611, 0.85035706, 0.04938295, -0.9328025, 0.13670611, 0.85035706, 0.04938295,
            -0.8231476, 0.1149258, 0.7187781, 0.14358711, -0.7598083, 0.0917983, 0.5723765, 0.13822867)
    out = np.zeros(2, np.float32)
    assert ork.ork_intersect_curve(p(f32(-2.2888184, 0.875, 3.0703347)), p(f32(0.51597244, -0.11002403, -0.6694812)), 0.0, 1e16, p(q), p(out)) == 0
    # rays along the tangent at a random parameter, passing 2.5 ... 6 radii from the axis: whatever is reported must lie on the tube
    rs = np.random.RandomState(9)
    ev = np.zeros(21, np.float32)
    reported = 0
    for _ in range(4000):
        q = np.concatenate([np.cumsum(rs.uniform(-0.2, 0.2, (4, 3)), 0), rs.uniform(0.03, 0.15, (4, 1))], 1).astype(np.float32).ravel()
        u = float(rs.choice([0.5, 0.25, 0.75, rs.uniform(0, 1)]))
        ork.ork_curve_eval(p(q), u, p(f32(0, 0, 0)), p(ev))
        c, tan = ev[:3].astype(np.float64), ev[4:7].astype(np.float64)
        if np.linalg.norm(tan) < 1e-6:
            continue
        tan /= np.linalg.norm(tan)
        side = np.cross(tan, rs.normal(size=3))
        side /= np.linalg.norm(side)
        o = (c + side * float(ev[3]) * rs.uniform(2.5, 6.0) - tan * 3.0).astype(np.float32)
        d = tan.astype(np.float32)
        if ork.ork_intersect_curve(p(o), p(d), 0.0, 1e16, p(q), p(out)):
            reported += 1
            ork.ork_curve_eval(p(q), float(out[1]), p(f32(0, 0, 0)), p(ev))
            hp = o.astype(np.float64) + float(out[0]) * d.astype(np.float64)
            assert abs(np.linalg.norm(hp - ev[:3]) - ev[3]) < 1e-3 + 0.02 * ev[3], (q, o, d, out)
    assert reported < 4000


@pytest.mark.parametrize("maker", ["cornell", "kitchen", "hair"])
def test_bvh_never_changes_a_result(maker):
    """closest hit and occlusion through the oracle's SAH BVH == brute force over every primitive, bit for bit
    (ties broken by the smaller (instance, primitive) key, conservative boxes)."""
    sc = {"cornell": scenes.cornell_box,
          "kitchen": lambda: scenes.kitchen_standin(seed=3, n_meshes=6, n_instances=61, tri_lo=40, tri_hi=300),
          "hair": lambda: scenes.hair_standin(seed=4, n_strands=300, n_cp=6)}[maker]()
    o = orklib.new_context()
    o.set_scene(sc.arrays())
    lo, hi = (-0.99, 0.99) if maker == "cornell" else ((-4.5, 4.5) if maker == "kitchen" else (-1.6, 1.6))
    rays = scenes.random_rays(4000, 5, lo, hi)
    a, b = o.trace(rays, 0, brute=True), o.trace(rays, 0, brute=False)
    for f in ("instance_id", "prim_id"):
        assert np.array_equal(a[f], b[f])
    for f in ("t", "u", "v"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32))
    rays["tmax"] = 1.0
    assert np.array_equal(o.trace(rays, 1, brute=True)["t"], o.trace(rays, 1, brute=False)["t"])
    assert (a["instance_id"] != 0xFFFFFFFF).mean() > 0.2


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_bake_world_definition(mode):
    """bake_world: a baked mesh / light-proxy instance is intersected in world space.  For every mode:
    the baked set follows the integer rule, the oracle's BVH still never changes a result, hits name the same (instance,
    primitive) as without baking, with t within rounding (t is the same parameter in both spaces) and barycentrics within rounding."""
    sc = scenes.kitchen_standin(seed=3, n_meshes=6, n_instances=61, tri_lo=40, tri_hi=300)
    arr = sc.arrays()
    ref = orklib.new_context()
    ref.set_bake(0)
    ref.set_scene(arr)
    o = orklib.new_context()
    o.set_bake(mode, 64)
    o.set_scene(arr)
    n = len(arr["instances"])
    baked = o.baked(n)
    inst, meshes = arr["instances"], arr["meshes"]
    valid = np.array([abs(np.linalg.det(t.reshape(3, 4)[:, :3].astype(np.float64))) > 0 for t in inst["transform"]])  # (the distant light's proxy is singular)
    users = np.bincount(inst["geom_id"][(inst["type"] != S.INSTANCE_CURVE) & valid], minlength=len(meshes))
    tris = meshes["index_count"] // 3
    geom, typ = inst["geom_id"], inst["type"]
    if mode == 4:  # the default: everything while the instanced triangles fit the budget (64 M), else mode 2
        assert tris[geom[valid & (typ != S.INSTANCE_CURVE)]].sum() <= 64e6
        mode = 3
    pick = np.array([bool(valid[i]) and (mode >= 3 or (mode >= 1 and users[geom[i]] == 1)) for i in range(n)])
    if mode == 2:  # small shared meshes only when no mesh instance stays behind; small light proxies are candidates on their own
        rest = valid & ~pick & (typ == S.INSTANCE_MESH)
        all_small = bool((tris[geom[rest]] <= 64).all())
        pick |= valid & (tris[geom] <= 64) & (all_small | (typ == S.INSTANCE_LIGHT))
    tlas_stays = bool((valid & ~pick & (inst["type"] == S.INSTANCE_MESH)).any())  # light proxies only follow when the top level empties
    for i in range(n):
        want = pick[i] and not (inst["type"][i] == S.INSTANCE_LIGHT and tlas_stays)
        assert bool(baked[i]) == bool(want), (i, mode)
    assert (mode == 0) == (baked.sum() == 0) and (mode < 3 or baked.sum() == valid.sum())
    rays = scenes.random_rays(6000, 9, -4.5, 4.5)
    a, b, r0 = o.trace(rays, 0, brute=True), o.trace(rays, 0, brute=False), ref.trace(rays, 0)
    for f in ("instance_id", "prim_id"):
        assert np.array_equal(a[f], b[f])
    for f in ("t", "u", "v"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32))
    same = (a["instance_id"] == r0["instance_id"]) & (a["prim_id"] == r0["prim_id"])
    # (coplanar contacts -- boards on walls, objects resting on boards -- are exact ties in t that rounding decides per space, and an
    # edge-on grazing hit may flip: whatever wins is at the same distance)
    assert same.mean() > 0.995
    both = ~same & (a["instance_id"] != 0xFFFFFFFF) & (r0["instance_id"] != 0xFFFFFFFF)
    assert np.allclose(a["t"][both], r0["t"][both], rtol=1e-4) and (~same & ~both).mean() < 1e-3
    hit = same & (a["instance_id"] != 0xFFFFFFFF)
    assert np.allclose(a["t"][hit], r0["t"][hit], rtol=2e-4) and np.allclose(a["u"][hit], r0["u"][hit], atol=2e-3) and np.allclose(a["v"][hit], r0["v"][hit], atol=2e-3)
    rays["tmax"] = 1.0
    s1, s0 = o.trace(rays, 1, brute=False)["t"], ref.trace(rays, 1)["t"]
    assert np.array_equal(o.trace(rays, 1, brute=True)["t"], s1) and (s1 == s0).mean() > 0.999


def test_instance_transform_and_masks():
    """A translated + scaled instance is hit where expected; light proxies are visible to radiance rays (mask 255)
    and invisible to shadow rays (RAY_MASK_SHADOW): OptixRenderParams.h:9-17."""
    sc = S.Scene()
    sc.addMaterial()
    vb, ib = S.deindex([(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)], [(0, 1, 2), (0, 2, 3)])
    m = sc.createMesh(vb, ib)
    sc.createInstance(S.INSTANCE_MESH, m, 0, S.translate((0, 0, -5)) @ S.scale((2, 2, 1)))
    sc.createLight({"type": 0, "xform": S.translate((0, 0, -2)), "useXform": True, "width": 1, "height": 1, "color": (1, 1, 1), "intensity": 1})
    o = orklib.new_context()
    o.set_scene(sc.arrays())
    r = np.zeros(3, S.RAY)
    r["origin"] = [(0, 0, 0), (1.5, 1.5, 0), (2.5, 0, 0)]
    r["dir"] = (0, 0, -1)
    r["tmax"] = 1e16
    h = o.trace(r, 0)
    assert h["instance_id"][0] == 1 and abs(h["t"][0] - 2) < 1e-6  # the light proxy is in front
    assert h["instance_id"][1] == 0 and abs(h["t"][1] - 5) < 1e-6  # scaled quad reaches +-2
    assert h["instance_id"][2] == 0xFFFFFFFF
    s = o.trace(r, 1)
    assert s["t"][0] > 0  # occluded by the quad; the light itself does not occlude
    r["tmax"] = 3.0
    assert o.trace(r, 1)["t"][0] < 0


def _sample(ork, mat, n3, k1, xi, inside=0):
    out = np.zeros(8, np.float32)
    ork.ork_bsdf_sample(p(mat), p(n3), p(n3), p(k1), p(xi), inside, p(out))
    return out


def _mat(typ, color=(0.8, 0.6, 0.4), rough=0.3, metallic=0.0, spec=0.5, ior=1.5):
    m = np.zeros((), S.MATERIAL)
    m["type"], m["base_color"], m["roughness"], m["metallic"], m["specular"], m["ior"] = typ, color, rough, metallic, spec, ior
    return m


@pytest.mark.parametrize("typ,metallic", [(0, 0.0), (1, 0.0), (1, 1.0)])  # (hair, type 3: tests/test_oracle_bsdf.py)
def test_bsdf_energy_and_pdf_consistency(ork, typ, metallic):
    """White-furnace style checks: E[bsdf_over_pdf] <= 1 per channel, sample.pdf == evaluate.pdf for the sampled
    direction, evaluate == pdf * bsdf_over_pdf, and the pdf integrates to <= 1 over the hemisphere."""
    rs = np.random.RandomState(7)
    mat = _mat(typ, color=(1, 1, 1), metallic=metallic, rough=0.35, spec=1.0)
    n = f32(0, 0, 1)
    k1 = f32(0.4, 0.1, 0.9)
    k1 /= np.linalg.norm(k1)
    acc = np.zeros(3)
    N = 4000
    for _ in range(N):
        xi = rs.rand(4).astype(np.float32)
        s = _sample(ork, mat, n, k1, xi)
        if s[7] == 0:
            continue
        acc += s[3:6]
        ev = np.zeros(7, np.float32)
        k2 = np.ascontiguousarray(s[:3])
        ork.ork_bsdf_evaluate(p(mat), p(n), p(n), p(k1), p(k2), p(ev))
        assert abs(ev[6] - s[6]) <= 2e-4 * max(1.0, s[6])
        assert np.allclose(ev[:3] + ev[3:6], s[6] * s[3:6], rtol=2e-3, atol=1e-5)
        assert abs(np.linalg.norm(k2) - 1) < 1e-5 and k2[2] > 0
    assert (acc / N <= 1.02).all() and (acc / N > 0.3).all()
    # pdf integrates to <= 1 (uniform hemisphere Monte Carlo)
    tot = 0.0
    M = 20000
    z = rs.rand(M)
    ph = rs.rand(M) * 2 * math.pi
    r = np.sqrt(1 - z * z)
    dirs = np.stack([r * np.cos(ph), r * np.sin(ph), z], 1).astype(np.float32)
    ev = np.zeros(7, np.float32)
    for d in dirs[:4000]:
        ork.ork_bsdf_evaluate(p(mat), p(n), p(n), p(k1), p(np.ascontiguousarray(d)), p(ev))
        tot += ev[6]
    integral = tot / 4000 * 2 * math.pi
    assert 0.85 < integral < 1.08


def test_glass_protocol(ork):
    """Specular events report pdf 0 (MDL convention the renderer relies on: closest_hit.cu:603); transmission flips
    sides; inside/outside selects ior1/ior2 (closest_hit.cu:496-498); total internal reflection."""
    mat = _mat(2, color=(0.9, 0.95, 1.0), rough=0.0, ior=1.5)  # clear glass (frosting_roughness 0); frosted: tests/test_oracle_bsdf.py
    n = f32(0, 0, 1)
    k1 = f32(0.0, 0.6, 0.8)
    refl = _sample(ork, mat, n, k1, f32(0.1, 0.2, 0.0, 0))  # xi.z = 0 < F -> reflection
    assert int(refl[7]) == (4 | 8) and refl[6] == 0 and np.allclose(refl[:3], (0, -0.6, 0.8), atol=1e-6)
    tr = _sample(ork, mat, n, k1, f32(0.1, 0.2, 0.99, 0))
    assert int(tr[7]) == (4 | 16) and tr[6] == 0 and tr[2] < 0
    assert abs(math.hypot(tr[0], tr[1]) - 0.6 / 1.5) < 1e-6  # Snell
    assert np.allclose(tr[3:6], (0.9, 0.95, 1.0))
    # from inside, beyond the critical angle: always reflect
    k1g = f32(0.0, 0.9, math.sqrt(1 - 0.81))
    tir = _sample(ork, mat, n, k1g, f32(0.1, 0.2, 0.99, 0), inside=1)
    assert int(tir[7]) == (4 | 8)
    ev = np.zeros(7, np.float32)
    ork.ork_bsdf_evaluate(p(mat), p(n), p(n), p(k1), p(f32(0, -0.6, 0.8)), p(ev))
    assert not ev.any()


def test_lambert_matches_the_metal_backend_semantics(ork):
    """pathtrace.metal:164-201: bsdf*cos = albedo*(n.k2)/pi, pdf = (n.k2)/pi, bsdf_over_pdf = albedo."""
    mat = _mat(0, color=(0.2, 0.5, 0.7))
    n = f32(0, 1, 0)
    k1 = f32(0.3, 0.8, 0.1)
    k1 /= np.linalg.norm(k1)
    k2 = f32(-0.5, 0.6, 0.2)
    k2 /= np.linalg.norm(k2)
    ev = np.zeros(7, np.float32)
    ork.ork_bsdf_evaluate(p(mat), p(n), p(n), p(k1), p(k2), p(ev))
    assert np.allclose(ev[:3], np.array([0.2, 0.5, 0.7]) * k2[1] / math.pi, rtol=1e-6)
    assert abs(ev[6] - k2[1] / math.pi) < 1e-7 and not ev[3:6].any()
    s = _sample(ork, mat, n, k1, f32(0.3, 0.6, 0, 0))
    assert np.allclose(s[3:6], (0.2, 0.5, 0.7)) and int(s[7]) == (1 | 8)
    # two-sided like MDL's libbsdf: viewed from the back, the sampled direction is on the viewer's side
    sb = _sample(ork, mat, n, -k1, f32(0.3, 0.6, 0, 0))
    assert sb[1] < 0


def test_offset_ray_branches(ork):
    """offset_ray (closest_hit.cu:218-233): integer-ULP offset away from the origin, fp offset near it."""
    out = np.zeros(3, np.float32)
    n = f32(0, 0, 1)
    ork.ork_offset_ray(p(f32(5, -7, 3)), p(n), p(out))
    assert out[0] == 5 and out[1] == -7 and out[2] > 3 and out[2] - 3 < 1e-4
    ork.ork_offset_ray(p(f32(5, -7, -3)), p(n), p(out))
    assert out[2] > -3  # negative coordinate: the integer is SUBTRACTED to move along +n
    ork.ork_offset_ray(p(f32(0.01, 0.0, 0.02)), p(n), p(out))
    assert abs(out[2] - (0.02 + 1.0 / 65536.0)) < 1e-9  # |p| < 1/32: float offset


def test_triangle_intersector_against_exact_rational_arithmetic(ork):
    """A8's triangle test has no reference arithmetic to compare with (OptiX is closed), so the oracle's watertight intersector is held
    against EXACT arithmetic: every float is a rational, so the ray / plane intersection, the barycentric weights and the inside / outside
    decision of random (ray, triangle) pairs are computed with `fractions.Fraction` -- no rounding at all.  Bars: the hit / miss decision
    agrees whenever the exact point is not within 1e-5 (relative to the triangle) of an edge or of the interval's ends; t, u, v agree with the
    exact values to 5e-7 x cond, cond = (1 + distance to the triangle / its shortest altitude) / cos(incidence): the vertices are subtracted from the
    origin in fp32 first, and a grazing ray multiplies every error; measured worst case 8.6e-8 x cond = 1.4 x 2^-24.  2 000 cases:
    random triangles of widely different size and position, rays aimed at, near and past them, both facings, non-unit directions."""
    from fractions import Fraction as F

    rs = np.random.RandomState(77)
    out = np.zeros(3, np.float32)
    decided = agreed = hits = 0
    worst = 0.0
    BAR = 5e-7
    for case in range(2000):
        scale = np.float32(10.0 ** rs.uniform(-2, 2))
        centre = (rs.normal(size=3) * 10.0 ** rs.uniform(-1, 1.5)).astype(np.float32)
        tri = (centre + rs.normal(size=(3, 3)).astype(np.float32) * scale).astype(np.float32)
        w = rs.dirichlet(np.ones(3)) if rs.rand() < 0.7 else rs.normal(size=3)  # inside 70 %, anywhere in the plane otherwise
        w = w / w.sum() if abs(w.sum()) > 1e-3 else np.array([0.2, 0.3, 0.5])
        target = (w[:, None] * tri.astype(np.float64)).sum(0)
        o = (target + rs.normal(size=3) * scale * 10.0 ** rs.uniform(-0.5, 1.5)).astype(np.float32)
        d = ((target - o.astype(np.float64)) * 10.0 ** rs.uniform(-1, 1)).astype(np.float32)
        if not np.any(d):
            continue
        tmin, tmax = 0.0, 1e16
        got = ork.ork_intersect_triangle(p(o), p(d), tmin, tmax, p(np.ascontiguousarray(tri.reshape(9))), p(out))
        # exact: P0 + u (P1 - P0) + v (P2 - P0) = O + t D
        P = [[F(float(x)) for x in v] for v in tri]
        O = [F(float(x)) for x in o]
        D = [F(float(x)) for x in d]
        e1 = [P[1][k] - P[0][k] for k in range(3)]
        e2 = [P[2][k] - P[0][k] for k in range(3)]

        def cross(a, b):
            return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]

        def dot(a, b):
            return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]

        pv = cross(D, e2)
        det = dot(e1, pv)
        if det == 0:
            assert got == 0  # exactly parallel: no hit
            continue
        tv = [O[k] - P[0][k] for k in range(3)]
        u = dot(tv, pv) / det
        qv = cross(tv, e1)
        v = dot(D, qv) / det
        t = dot(e2, qv) / det
        margin = min(u, v, 1 - u - v)  # > 0 inside
        clear = abs(float(margin)) > 1e-5 and abs(float(t)) > 1e-5 * (1 + abs(float(t)))
        if not clear:
            continue
        decided += 1
        want = 1 if (margin > 0 and t > tmin and t <= F(tmax)) else 0
        assert got == want, (case, float(u), float(v), float(t), got)
        agreed += 1
        if want:
            hits += 1
            # conditioning: the vertices are subtracted from the origin in fp32 first, so the edge functions see the triangle with an absolute
            # error of ~2^-24 x (distance to the triangle), i.e. relative to its own size (for barycentrics: its shortest altitude): 2^-24 x distance / size
            far = max(abs(float(P[i][k] - O[k])) for i in range(3) for k in range(3))
            nrm = cross(e1, e2)
            size = math.sqrt(float(dot(nrm, nrm))) / max(math.sqrt(float(dot(e, e))) for e in (e1, e2, [e2[k] - e1[k] for k in range(3)]))  # shortest altitude
            cosi = abs(float(dot(D, nrm))) / math.sqrt(float(dot(D, D)) * float(dot(nrm, nrm)))  # a grazing ray multiplies every error by 1 / cos
            cond = (1.0 + far / size) / cosi
            worst = max(worst, max(abs(out[1] - float(u)), abs(out[2] - float(v)), abs(out[0] / float(t) - 1.0)) / cond)
            assert abs(out[0] - float(t)) <= BAR * cond * abs(float(t)), (case, out[0], float(t), cond)
            assert abs(out[1] - float(u)) <= BAR * cond and abs(out[2] - float(v)) <= BAR * cond, (case, out[1:], float(u), float(v), cond)
    assert decided > 1800 and hits > 1000 and agreed == decided
    assert worst < 2e-7  # (measured 8.6e-8 = 1.4 x 2^-24 of the conditioned quantity)


def test_curve_intersector_finds_the_first_entry_into_the_swept_volume(ork):
    """A6 / A8, curves: OptiX's round cubic B-spline primitive is closed, so the oracle's phantom intersector (`ork_intersect_curve`) is held
    against a brute-force fp64 statement of what it must find: the FIRST point along the ray that lies inside the union of the spheres
    (C(u), r(u)), u in [0, 1] -- located by a dense scan (1 500 ray steps x 2 001 curve parameters, then 40 bisections).  Rays are aimed at the
    middle of the segment (u0 in [0.15, 0.85]) from outside, at a random offset from the axis: those that pass within 0.8 r must hit at the scan's
    t (bar 3e-4, the on-surface test's; measured worst 1.7e-5), those that pass farther than 1.25 r from every point of the curve must miss."""
    rs = np.random.RandomState(11)
    q = f32(0, 0, 0, 0.05, 1, 0.3, 0, 0.06, 2, -0.2, 0.4, 0.04, 3, 0, 0, 0.03)
    cp = q.reshape(4, 4).astype(np.float64)
    us = np.linspace(0.0, 1.0, 2001)
    # uniform cubic B-spline basis (the published one; tests/test_oracle_golden.py pins ork_curve_eval to it)
    B = np.stack([(1 - us) ** 3, 3 * us ** 3 - 6 * us ** 2 + 4, -3 * us ** 3 + 3 * us ** 2 + 3 * us + 1, us ** 3], axis=1) / 6.0
    CU = B @ cp  # [2001, 4]: centre xyz, radius

    def depth(P):  # P [n, 3] -> min over u of |P - C(u)| - r(u)   (< 0 inside the swept volume)
        dist = np.linalg.norm(P[:, None, :] - CU[None, :, :3], axis=2) - CU[None, :, 3]
        return dist.min(axis=1)

    out = np.zeros(2, np.float32)
    hits = misses = 0
    worst = 0.0
    for _ in range(110):
        k = rs.randint(300, 1700)
        c, r = CU[k, :3], CU[k, 3]
        off = rs.normal(size=3)
        off *= rs.choice([rs.uniform(0.0, 0.8), rs.uniform(1.6, 3.0)]) * r / np.linalg.norm(off)
        target = c + off
        o = (target + rs.normal(size=3) * 1.5).astype(np.float32)
        d = target - o.astype(np.float64)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        o64, d64 = o.astype(np.float64), d.astype(np.float64)
        ts = np.linspace(0.0, 6.0, 1501)
        dep = depth(o64[None, :] + ts[:, None] * d64[None, :])
        if dep[0] < 0:
            continue  # (origin inside the tube: not this test's case)
        got = ork.ork_intersect_curve(p(o), p(d), 0.0, 1e16, p(q), p(out))
        inside = np.nonzero(dep < 0)[0]
        if len(inside) == 0:
            if dep.min() > 0.25 * CU[:, 3].min():
                assert got == 0, (o, d, dep.min())
                misses += 1
            continue
        lo, hi = ts[inside[0] - 1], ts[inside[0]]
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            if depth((o64 + mid * d64)[None, :])[0] < 0:
                hi = mid
            else:
                lo = mid
        t_ref = 0.5 * (lo + hi)
        # entry through the side of the tube, not through the opening of its uncapped ends
        P = o64 + t_ref * d64
        u_ref = us[np.argmin(np.linalg.norm(P[None, :] - CU[:, :3], axis=1) - CU[:, 3])]
        if u_ref < 0.02 or u_ref > 0.98 or dep.min() > -0.2 * r:
            continue
        assert got == 1, (o, d, t_ref, u_ref)
        worst = max(worst, abs(float(out[0]) - t_ref))
        assert abs(float(out[0]) - t_ref) < 3e-4, (float(out[0]), t_ref, u_ref, float(out[1]))
        assert abs(float(out[1]) - u_ref) < 5e-3
        hits += 1
    assert hits > 35 and misses > 20, (hits, misses, worst)


def test_world_curves_rule_decides_the_light_proxies():
    """"World curves" (round 5; the table holds two trees since round 6, and curve instances under bit-exact identity transforms share ONE of them):
    when the scene's curve instances need at most two trees and no mesh or light instance is left unbaked, the curve sets need no top level --
    the product walks their trees from the world-only kernel, each instance's transform applied to the ray as at a TLAS leaf -- and the light
    proxies follow the meshes into world space.  The rule is an integer rule both sides evaluate; here the checker's side: one curve instance,
    under any transform -> lights baked; 16 or 17 copies under the identity (one merged tree) -> baked; the merged tree + one moved instance, or
    two moved instances (two trees) -> baked; five instances under ONE non-identity transform (one merged tree) -> baked; three instances under transforms of their own, or bake mode 1 (the shared light quad stays
    behind) -> not.  The curves' own hit records never depend on it: closest hits equal the unbaked context's."""
    from tests.test_gpu_parity import camera_rays

    sc = scenes.hair_standin(seed=5, n_strands=4000, n_cp=7)
    arr = dict(sc.arrays())
    inst = arr["instances"]
    lights = np.nonzero(inst["type"] == S.INSTANCE_LIGHT)[0]
    curves = np.nonzero(inst["type"] == S.INSTANCE_CURVE)[0]
    assert len(lights) == 2 and len(curves) == 1

    def baked_lights(a, mode=4):
        o = orklib.new_context()
        o.set_bake(mode)
        o.set_scene(a)
        return o, [int(v) for v in o.baked(len(a["instances"]))[lights]]

    o_id, b = baked_lights(arr)
    assert b == [1, 1]
    rays = np.concatenate([camera_rays(sc, 48, 48, 20000, 3), scenes.random_rays(20000, 4, -1.5, 1.5)])
    for change, want in (("rotate", [1, 1]), ("mode1", [0, 0]), ("many", [1, 1]), ("sixteen", [1, 1]), ("merged+1", [1, 1]), ("2 moved", [1, 1]), ("3 moved", [0, 0]), ("5 under one transform", [1, 1])):
        a2 = dict(arr)
        i2 = inst.copy()
        mode = 4
        if change == "rotate":
            m = np.eye(4)
            m[:3] = i2["transform"][curves[0]].reshape(3, 4)
            i2["transform"][curves[0]] = (S.translate((0.1, -0.2, 0.05)) @ m @ S.rotate((0, 1, 0), 0.3) @ S.scale((1.0, 0.7, 1.3)))[:3].astype(np.float32).reshape(12)
        elif change == "mode1":
            mode = 1
        elif change == "5 under one transform":
            M = (S.translate((0.1, 0.0, -0.05)) @ S.rotate((0, 1, 0), 0.4))[:3].astype(np.float32).reshape(12)
            i2 = np.concatenate([i2, np.repeat(i2[curves[:1]], 4)])
            i2["transform"][i2["type"] == S.INSTANCE_CURVE] = M
        elif change in ("many", "sixteen"):
            i2 = np.concatenate([i2, np.repeat(i2[curves[:1]], 16 if change == "many" else 15)])
        else:
            extra = np.repeat(i2[curves[:1]], 3 if change == "merged+1" else (1 if change == "2 moved" else 2))
            moved = {"merged+1": [0], "2 moved": [0], "3 moved": [0, 1]}[change]
            for j in moved:
                extra["transform"][j] = S.translate((0.01 * (j + 1), 0.0, 0.0))[:3].astype(np.float32).reshape(12)
            if change != "merged+1":  # ... and the original instance moves too
                i2["transform"][curves[0]] = S.translate((0.0, 0.02, 0.0))[:3].astype(np.float32).reshape(12)
            i2 = np.concatenate([i2, extra])
        a2["instances"] = i2
        o2, b2 = baked_lights(a2, mode)
        assert b2 == want, change
        if change == "rotate":  # the transformed curve set's hit records with and without the rule's consequences
            ref = orklib.new_context()
            ref.set_bake(0)
            ref.set_scene(a2)
            a, r0 = o2.trace(rays, 0), ref.trace(rays, 0)
            on_curves = np.isin(r0["instance_id"], curves)
            assert on_curves.sum() > 200
            for f in ("instance_id", "prim_id"):
                assert np.array_equal(a[f][on_curves], r0[f][on_curves])
            for f in ("t", "u", "v"):
                assert np.array_equal(a[f][on_curves].view(np.uint32), r0[f][on_curves].view(np.uint32))
