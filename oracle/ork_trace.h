// ORACLE -- TEST INFRASTRUCTURE ONLY (see ork_math.h).
//
// ork_trace.h: what the reference delegates to closed NVIDIA OptiX 8:
// two-level BVH (per-mesh / per-curve-set BLAS + TLAS over instances), closest-hit and any-hit queries,
// ray/triangle and ray/round-cubic-B-spline intersection.  Call sites restated:
//   optixAccelBuild GAS/mesh  src/render/optix/OptixRender.cpp:318-386   (float3 verts stride 32, uint3 indices)
//   optixAccelBuild GAS/curve src/render/optix/OptixRender.cpp:218-316   (ROUND_CUBIC_BSPLINE, end caps off)
//   optixAccelBuild IAS       src/render/optix/OptixRender.cpp:412-495   (visibility masks 1/2/4)
//   optixTrace radiance       src/render/optix/OptixRender.cu:120-129    (mask 255, closest hit)
//   optixTrace occlusion      src/render/optix/OptixRender_radiance_closest_hit.cu:185-197 (mask 3, first hit)
// OptiX's arithmetic is not in the reference tree (binary driver component) => PARITY UNPINNED for this file:
// the contract is  HIP kernel == this file  (bit-exact instance/primitive ids, t/u/v), plus analytic
// known-answer tests (tests/test_oracle_intersect.py).
//
// The intersection routines are specified so that the RESULT does not depend on the acceleration
// structure: closest hit = min over primitives of t, ties broken by the smaller (instance, primitive)
// key; box tests are conservative.  That lets a brute-force loop, this file's SAH BVH and the GPU's LBVH
// be compared bit for bit.
#pragma once
#include "ork_core.h"
#include <algorithm>
#include <vector>

namespace ork
{

struct Ray
{
    f3 o;
    float tmin;
    f3 d;
    float tmax;
};

struct Hit
{
    float t;
    uint32_t inst;
    uint32_t prim;
    float u, v;
};

// ---------------------------------------------------------------------------------------------
// Ray / triangle: watertight edge-function test (Woop, Benthin, Wald 2013), single precision, with the paper's
// fp64 re-evaluation of an edge function that rounds to zero, and a rejection of depths within rounding noise of tmin.  Barycentrics follow optixGetTriangleBarycentrics: hit = (1-u-v) p0 + u p1 + v p2.
// Accepts tmin < t <= tmax; the caller resolves ties at t == best by the (instance, primitive) key and
// starts from best = ray.tmax with key 0, which makes the ray interval open at both ends.
// ---------------------------------------------------------------------------------------------
struct RayShear
{
    int kx, ky, kz;
    float Sx, Sy, Sz;
};
static inline float comp(const f3& v, int k)
{
    return k == 0 ? v.x : (k == 1 ? v.y : v.z);
}
static inline RayShear make_shear(const f3& d)
{
    RayShear s;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    s.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    s.kx = s.kz + 1;
    if (s.kx == 3)
        s.kx = 0;
    s.ky = s.kx + 1;
    if (s.ky == 3)
        s.ky = 0;
    if (comp(d, s.kz) < 0.0f)
    {
        const int t = s.kx;
        s.kx = s.ky;
        s.ky = t;
    }
    const float dz = comp(d, s.kz);
    s.Sz = 1.0f / dz; // one division; the shear factors use the reciprocal (own definition, identical on the GPU)
    s.Sx = comp(d, s.kx) * s.Sz;
    s.Sy = comp(d, s.ky) * s.Sz;
    return s;
}
static inline bool intersect_triangle(const f3& o, const RayShear& s, float tmin, float tmax, const f3& p0, const f3& p1,
                                      const f3& p2, float& t_out, float& u_out, float& v_out)
{
    const f3 A = p0 - o, B = p1 - o, C = p2 - o;
    const float Akz = comp(A, s.kz), Bkz = comp(B, s.kz), Ckz = comp(C, s.kz);
    const float Ax = comp(A, s.kx) - s.Sx * Akz;
    const float Ay = comp(A, s.ky) - s.Sy * Akz;
    const float Bx = comp(B, s.kx) - s.Sx * Bkz;
    const float By = comp(B, s.ky) - s.Sy * Bkz;
    const float Cx = comp(C, s.kx) - s.Sx * Ckz;
    const float Cy = comp(C, s.ky) - s.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    // >= the sum of the absolute values of the six products above: bounds the cancellation in U, V, W (used below)
    const float S = ((fabsf(Ax) + fabsf(Bx)) + fabsf(Cx)) * ((fabsf(Ay) + fabsf(By)) + fabsf(Cy));
    if (U == 0.0f || V == 0.0f || W == 0.0f)
    {
        // Woop et al.'s fallback: an edge function that rounds to zero is re-evaluated in fp64, where the products of two
        // floats are exact and the sign of their difference is therefore exact.  Without it a triangle seen exactly edge-on
        // (projected vertices collinear with the ray) passes the sign test on rounding noise and reports a "hit" far outside
        // its own bounding box -- which conservative box tests cull, i.e. the result would depend on the hierarchy.
        const double Ud = (double)Cx * (double)By - (double)Cy * (double)Bx;
        const double Vd = (double)Ax * (double)Cy - (double)Ay * (double)Cx;
        const double Wd = (double)Bx * (double)Ay - (double)By * (double)Ax;
        if ((Ud < 0.0 || Vd < 0.0 || Wd < 0.0) && (Ud > 0.0 || Vd > 0.0 || Wd > 0.0))
            return false;
        U = (float)Ud;
        V = (float)Vd;
        W = (float)Wd;
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
        return false;
    const float det = (U + V) + W;
    if (det == 0.0f)
        return false;
    const float Az = s.Sz * Akz, Bz = s.Sz * Bkz, Cz = s.Sz * Ckz;
    const float T = (U * Az + V * Bz) + W * Cz;
    // A depth closer to the start of the ray than the rounding noise of its own evaluation is rejected: the sign of
    // t - tmin would be arbitrary there (an origin lying on the triangle; worst for sliver triangles, whose barycentric weights
    // are ill-conditioned), while the boxes around the triangle decide by geometry -- the hit would exist in one hierarchy and
    // not in another.  S bounds the cancellation in the edge functions; for a well-shaped triangle the threshold is a few
    // 2^-20 of the parametric distance to the farthest vertex.
    const float noise = 0x1p-20f * (S * fmaxf(fmaxf(fabsf(Az), fabsf(Bz)), fabsf(Cz)));
    const float q = T - tmin * det;
    if (!((det > 0.0f ? q : -q) > noise))
        return false;
    const float rcpDet = 1.0f / det;
    const float t = T * rcpDet;
    if (!(t > tmin && t <= tmax))
        return false;
    t_out = t;
    u_out = V * rcpDet;
    v_out = W * rcpDet;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Ray / round cubic B-spline segment with varying radius, end caps off.
// Method: Reshetov & Luebke, "Phantom Ray-Hair Intersector" (HPG 2018): iterate on the curve parameter,
// intersecting the ray with the cone tangent to the swept surface at the current parameter; regula falsi
// with a bisection every 4th step; stop at |dt| < 5e-5; both ends are tried.  Own tolerances (OptiX's
// built-in intersector is closed).  Works in ray-centric coordinates with a unit direction; the returned t
// is in units of the (possibly non-unit) input direction.
// ---------------------------------------------------------------------------------------------
static inline void onb_from_z(const f3& n, f3& b1, f3& b2) // Duff et al. 2017, branchless ONB
{
    const float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    b1 = f3{ 1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x };
    b2 = f3{ b, sign + n.y * n.y * a, -n.y };
}
struct ConeIsect
{
    f3 c0, cd;
    float s, dt, dp, dc, sp, dd; // (dd = |cd|^2)
    bool intersect(float r, float dr)
    {
        const float r2 = r * r;
        const float drr = r * dr;
        float ddd = cd.x * cd.x + cd.y * cd.y;
        dp = c0.x * c0.x + c0.y * c0.y;
        const float cdd = c0.x * cd.x + c0.y * cd.y;
        const float cxd = c0.x * cd.y - c0.y * cd.x;
        const float c = ddd;
        const float b = cd.z * (drr - cdd);
        const float cdz2 = cd.z * cd.z;
        ddd += cdz2;
        const float a = ((2.0f * drr * cdd + cxd * cxd) - ddd * r2) + dp * cdz2;
        const float det = b * b - a * c;
        s = (b - (det > 0.0f ? sqrtf(det) : 0.0f)) / c;
        dt = (s * cd.z - cdd) / ddd;
        dd = ddd;
        dc = s * s + dp;
        sp = cdd / cd.z;
        dp += sp * sp;
        return det > 0.0f;
    }
};
// q: 4 B-spline control points (xyz, radius) in the ray's object space.
static inline bool intersect_curve_segment(const f3& o, const f3& d, float tmin, float tmax, const f4* q, float& t_out,
                                           float& u_out)
{
    const float dlen = sqrtf(dot(d, d));
    const float inv_dlen = 1.0f / dlen;
    const f3 dn = d * inv_dlen;
    f3 bx, by;
    onb_from_z(dn, bx, by);
    // control points to ray-centric coordinates
    f4 qc[4];
    for (int i = 0; i < 4; ++i)
    {
        const f3 p = mk3(q[i]) - o;
        qc[i] = f4{ dot(p, bx), dot(p, by), dot(p, dn), q[i].w };
    }
    CubicInterpolator poly;
    poly.initializeFromBSpline(qc);
    // end points of the segment (u = 0, u = 1) decide which end to start from
    const f4 e0 = poly.position4(0.0f);
    const f4 e1 = poly.position4(1.0f);
    float tstart = (e1.z - e0.z) > 0.0f ? 0.0f : 1.0f;
    // Both ends are always tried and the NEARER accepted root wins.  The roots the two passes converge to do not depend on
    // [tmin, tmax]; returning the first accepted one would make the closest hit depend on the order in which segments are
    // visited (a first pass that converges to the far side of a thick tube is accepted under tmax = 1e16 and rejected
    // once a nearer hit elsewhere has shrunk the interval).
    bool found = false;
    for (int ep = 0; ep < 2; ++ep)
    {
        float t = tstart;
        ConeIsect rci;
        float told = 0.0f, dt1 = 0.0f, dt2 = 0.0f;
        for (int i = 0; i < 40; ++i)
        {
            const f4 c4 = poly.position4(t);
            // derivative without the reference's triple-knot nudge: plain Horner derivative
            const f4 d4 = ((3.0f * poly.p[0] * t) + 2.0f * poly.p[1]) * t + poly.p[2];
            rci.c0 = mk3(c4);
            rci.cd = mk3(d4);
            const bool phantom = !rci.intersect(c4.w, d4.w);
            if (!phantom && fabsf(rci.dt) < 5e-5f)
            {
                const float s = (rci.s + rci.c0.z) * inv_dlen;
                // A converged point lies ON the tube: its distance from the curve point, less the part along the tangent (dt |c'|), is the radius there.  A ray
                // (nearly) parallel to the tangent makes the cone's quadratic degenerate -- c = |c'_xy|^2 -> 0, b - sqrt(det) cancels to 0, dt comes out small and
                // det > 0 by rounding -- and the iteration "converges" at once on a point half a tube length away (round 6, fuzz_render seed 5483: 0.49 from a curve
                // point of radius 0.096).  Such a root is not a hit; every bound the hierarchies keep (hull + largest radius) relies on that: the radius along the
                // segment stays inside the hull of the control radii, and an accepted point is within 1.0005 of the cone's radius of the curve point.
                const float radial2 = rci.dc - (rci.dt * rci.dt) * rci.dd;
                const float rb = c4.w + d4.w * rci.dt; // the cone's radius at that offset along the tangent (linear: exact for the cone)
                if (radial2 <= 1.001f * (rb * rb) && s > tmin && s <= tmax && t >= 0.0f && t <= 1.0f && (!found || s < t_out))
                {
                    t_out = s;
                    u_out = t;
                    found = true;
                }
                break; // converged: try the other end
            }
            // A run that has converged onto a point the ray does NOT touch (phantom and |dt| below the tolerance: the closest approach of
            // a miss) is over: the next steps repeat that point until the cap.  Measured on 1.6 M random runs (round 3, thin hair and
            // thick tubes): 61 % of the thin-hair runs end up there, none of them ever turned into a hit, and stopping here changed no
            // result -- but the mean run went from 25.7 to 3.7 steps.
            if (phantom && fabsf(rci.dt) < 5e-5f)
                break;
            rci.dt = fminf(rci.dt, 0.5f);
            rci.dt = fmaxf(rci.dt, -0.5f);
            dt1 = dt2;
            dt2 = rci.dt;
            if (dt1 * dt2 < 0.0f)
            {
                float tnext;
                if ((i & 3) == 0)
                    tnext = 0.5f * (told + t);
                else
                    tnext = (dt2 * told - dt1 * t) / (dt2 - dt1);
                told = t;
                t = tnext;
            }
            else
            {
                told = t;
                t += rci.dt;
            }
            if (!(t >= 0.0f && t <= 1.0f))
                break;
        }
        tstart = 1.0f - tstart;
    }
    return found;
}

// ---------------------------------------------------------------------------------------------
// BVH (binned SAH, binary).  Conservative slab test: tFar is padded by 1 + 2^-22, and leaf boxes are
// inflated by 2^-20 of their largest absolute coordinate, so that the box test never rejects a primitive
// the primitive test would accept.
// ---------------------------------------------------------------------------------------------
struct Aabb
{
    f3 lo, hi;
    void reset()
    {
        lo = mk3(INFINITY);
        hi = mk3(-INFINITY);
    }
    void grow(const f3& p)
    {
        lo = f3{ fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z) };
        hi = f3{ fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z) };
    }
    void grow(const Aabb& b)
    {
        grow(b.lo);
        grow(b.hi);
    }
    float half_area() const
    {
        const f3 e = hi - lo;
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
};
static inline void inflate(Aabb& b)
{
    const float m = fmaxf(fmaxf(fmaxf(fabsf(b.lo.x), fabsf(b.lo.y)), fmaxf(fabsf(b.lo.z), fabsf(b.hi.x))),
                          fmaxf(fabsf(b.hi.y), fabsf(b.hi.z)));
    const float e = m * 0x1p-20f + 1e-30f;
    b.lo = b.lo - mk3(e);
    b.hi = b.hi + mk3(e);
}
static inline bool slab(const Aabb& b, const f3& o, const f3& inv, float tmin, float tmax)
{
    // The subtraction (plane - o) rounds at the magnitude of the LARGER operand: with the ray origin far from a small box (object
    // space of a strongly scaled instance: |o| = |R^-1| |o_world - T|) that absolute error exceeds the 2^-20 relative inflation of the
    // box and a primitive the brute-force loop accepts could be culled (tools/fuzz_hits.py seed 1735, round 2: 1 ray in 10^8; the
    // product's quantised boxes are at least one cell thick and were right).  Each plane is therefore moved outward by
    // (|plane| + |o|) * 2^-16 before the subtraction -- which also covers the noise of the triangle test itself, whose edge
    // functions are differences of coordinates relative to the origin (a ray aimed at a vertex from 10^4 object sizes away "hits" a
    // triangle it passes 10^-3 sizes beside: the brute-force loop reports that hit, so the boxes must let the ray through);
    // tFar keeps its 1 + 2^-20 for the rounding of inv and of the products.
    const float px = (fmaxf(fabsf(b.lo.x), fabsf(b.hi.x)) + fabsf(o.x)) * 0x1p-16f;
    const float py = (fmaxf(fabsf(b.lo.y), fabsf(b.hi.y)) + fabsf(o.y)) * 0x1p-16f;
    const float pz = (fmaxf(fabsf(b.lo.z), fabsf(b.hi.z)) + fabsf(o.z)) * 0x1p-16f;
    float t0x = ((b.lo.x - px) - o.x) * inv.x, t1x = ((b.hi.x + px) - o.x) * inv.x;
    float t0y = ((b.lo.y - py) - o.y) * inv.y, t1y = ((b.hi.y + py) - o.y) * inv.y;
    float t0z = ((b.lo.z - pz) - o.z) * inv.z, t1z = ((b.hi.z + pz) - o.z) * inv.z;
    // NaN-safe ordering (0 * inf): fminf/fmaxf drop NaNs
    const float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    return tn <= tf * 1.00000095367431640625f; // 1 + 2^-20
}

struct BvhNode
{
    Aabb box;
    uint32_t left; // internal: index of left child (right = left + 1); leaf: first primitive
    uint32_t count; // 0 = internal
};

struct Bvh
{
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> prim; // permutation
    Aabb bounds() const
    {
        return nodes.empty() ? Aabb{ mk3(0), mk3(0) } : nodes[0].box;
    }

    void build(const std::vector<Aabb>& boxes, uint32_t leafMax)
    {
        const uint32_t n = (uint32_t)boxes.size();
        prim.resize(n);
        for (uint32_t i = 0; i < n; ++i)
            prim[i] = i;
        nodes.clear();
        if (n == 0)
            return;
        nodes.reserve(2 * n);
        std::vector<f3> cent(n);
        for (uint32_t i = 0; i < n; ++i)
            cent[i] = (boxes[i].lo + boxes[i].hi) * 0.5f;
        nodes.push_back(BvhNode{});
        struct Item
        {
            uint32_t node, first, count;
        };
        std::vector<Item> stack;
        stack.push_back(Item{ 0, 0, n });
        while (!stack.empty())
        {
            const Item it = stack.back();
            stack.pop_back();
            Aabb nb, cb;
            nb.reset();
            cb.reset();
            for (uint32_t i = it.first; i < it.first + it.count; ++i)
            {
                nb.grow(boxes[prim[i]]);
                cb.grow(cent[prim[i]]);
            }
            nodes[it.node].box = nb;
            if (it.count <= leafMax)
            {
                nodes[it.node].left = it.first;
                nodes[it.node].count = it.count;
                continue;
            }
            // binned SAH over the largest centroid axis, 16 bins
            const f3 ce = cb.hi - cb.lo;
            int axis = 0;
            if (ce.y > ce.x)
                axis = 1;
            if (ce.z > comp(ce, axis))
                axis = 2;
            const float cmin = comp(cb.lo, axis), cext = comp(ce, axis);
            uint32_t mid = it.first + it.count / 2;
            bool split_done = false;
            if (cext > 0.0f)
            {
                const int NB = 16;
                Aabb bb[NB];
                uint32_t bc[NB];
                for (int b = 0; b < NB; ++b)
                {
                    bb[b].reset();
                    bc[b] = 0;
                }
                const float scale = (float)NB / cext;
                auto bin_of = [&](uint32_t p) {
                    int b = (int)((comp(cent[p], axis) - cmin) * scale);
                    return b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                };
                for (uint32_t i = it.first; i < it.first + it.count; ++i)
                {
                    const int b = bin_of(prim[i]);
                    bb[b].grow(boxes[prim[i]]);
                    bc[b]++;
                }
                float rightA[NB];
                uint32_t rightC[NB];
                Aabb acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = NB - 1; b > 0; --b)
                {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    rightA[b] = cnt ? acc.half_area() : 0.0f;
                    rightC[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                float best = INFINITY;
                int bestB = -1;
                for (int b = 0; b < NB - 1; ++b)
                {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    if (cnt == 0 || rightC[b + 1] == 0)
                        continue;
                    const float cost = acc.half_area() * (float)cnt + rightA[b + 1] * (float)rightC[b + 1];
                    if (cost < best)
                    {
                        best = cost;
                        bestB = b;
                    }
                }
                if (bestB >= 0)
                {
                    uint32_t* first = prim.data() + it.first;
                    uint32_t* last = first + it.count;
                    uint32_t* m = std::partition(first, last, [&](uint32_t p) { return bin_of(p) <= bestB; });
                    mid = (uint32_t)(m - prim.data());
                    split_done = (mid > it.first && mid < it.first + it.count);
                }
            }
            if (!split_done)
            {
                mid = it.first + it.count / 2;
                std::nth_element(prim.begin() + it.first, prim.begin() + mid, prim.begin() + it.first + it.count,
                                 [&](uint32_t a, uint32_t b) { return comp(cent[a], axis) < comp(cent[b], axis); });
            }
            const uint32_t l = (uint32_t)nodes.size();
            nodes.push_back(BvhNode{});
            nodes.push_back(BvhNode{});
            nodes[it.node].left = l;
            nodes[it.node].count = 0;
            stack.push_back(Item{ l + 1, mid, it.first + it.count - mid });
            stack.push_back(Item{ l, it.first, mid - it.first });
        }
        // conservative inflation of every box (cheap and safe)
        for (auto& nd : nodes)
            inflate(nd.box);
    }
};

} // namespace ork
