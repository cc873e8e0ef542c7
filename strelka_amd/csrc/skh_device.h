// strelka_hip -- device-side arithmetic of the MI355X wavefront path tracer (gfx950 only).
//
// Everything here is __device__ code used by the kernels in skh_kernels.h.  The arithmetic of the pieces that exist
// in the reference follows it operation for operation (file:line cited per function, paths relative to
// arhix52/Strelka); the pieces the reference delegates to closed OptiX / MDL SDK (intersection, BSDF) are this
// project's own definitions, documented in DESIGN.md.  Build with -ffp-contract=off: rounding must not depend on
// FMA contraction (parity with the CPU oracle is checked bit for bit where the arithmetic is + - * / sqrt only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "skh_libm.h"

namespace skh
{

struct v3
{
    float x, y, z;
};
struct v4
{
    float x, y, z, w;
};

#define SKH_DI static __device__ __forceinline__

SKH_DI v3 mk3(float x, float y, float z)
{
    v3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
SKH_DI v3 mk3(float s)
{
    return mk3(s, s, s);
}
SKH_DI v3 mk3(const v4& a)
{
    return mk3(a.x, a.y, a.z);
}
SKH_DI v3 mk3(const float4& a)
{
    return mk3(a.x, a.y, a.z);
}
SKH_DI v4 mk4(float x, float y, float z, float w)
{
    v4 r;
    r.x = x;
    r.y = y;
    r.z = z;
    r.w = w;
    return r;
}
SKH_DI v3 operator+(const v3& a, const v3& b)
{
    return mk3(a.x + b.x, a.y + b.y, a.z + b.z);
}
SKH_DI v3 operator-(const v3& a, const v3& b)
{
    return mk3(a.x - b.x, a.y - b.y, a.z - b.z);
}
SKH_DI v3 operator-(const v3& a)
{
    return mk3(-a.x, -a.y, -a.z);
}
SKH_DI v3 operator*(const v3& a, const v3& b)
{
    return mk3(a.x * b.x, a.y * b.y, a.z * b.z);
}
SKH_DI v3 operator*(const v3& a, float s)
{
    return mk3(a.x * s, a.y * s, a.z * s);
}
SKH_DI v3 operator*(float s, const v3& a)
{
    return mk3(s * a.x, s * a.y, s * a.z);
}
SKH_DI v3 operator/(const v3& a, const v3& b)
{
    return mk3(a.x / b.x, a.y / b.y, a.z / b.z);
}
// sutil semantics: vector / scalar multiplies by the reciprocal (sutil/vec_math.h:487-491)
SKH_DI v3 operator/(const v3& a, float s)
{
    const float inv = 1.0f / s;
    return a * inv;
}
SKH_DI v4 operator+(const v4& a, const v4& b)
{
    return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
SKH_DI v4 operator*(const v4& a, float s)
{
    return mk4(a.x * s, a.y * s, a.z * s, a.w * s);
}
SKH_DI v4 operator*(float s, const v4& a)
{
    return mk4(s * a.x, s * a.y, s * a.z, s * a.w);
}
SKH_DI v4 operator/(const v4& a, float s)
{
    const float inv = 1.0f / s;
    return a * inv;
}
SKH_DI float dot(const v3& a, const v3& b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
SKH_DI v3 cross(const v3& a, const v3& b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
SKH_DI float length(const v3& v)
{
    return sqrtf(dot(v, v));
}
SKH_DI v3 normalize(const v3& v) // sutil/vec_math.h:549-553
{
    const float invLen = 1.0f / sqrtf(dot(v, v));
    return v * invLen;
}
SKH_DI float clampf(float f, float a, float b) // sutil/vec_math.h:123-126
{
    return fmaxf(a, fminf(f, b));
}
SKH_DI float saturatef(float v)
{
    return clampf(v, 0.0f, 1.0f);
}
SKH_DI v3 lerp3(const v3& a, const v3& b, float t) // sutil/vec_math.h:504-507
{
    return a + t * (b - a);
}
SKH_DI bool all3(const v3& v) // sutil/vec_math_adv.h:39-42
{
    return v.x != 0.0f && v.y != 0.0f && v.z != 0.0f;
}
SKH_DI bool isnan3(const v3& v)
{
    return isnan(v.x) || isnan(v.y) || isnan(v.z);
}
SKH_DI float comp(const v3& v, int k)
{
    return k == 0 ? v.x : (k == 1 ? v.y : v.z);
}

// row-major 4x4 * float4: sutil/Matrix.h:467-488
SKH_DI v4 mul44(const float* m, const v4& v)
{
    v4 r;
    r.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w;
    r.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * v.w;
    r.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * v.w;
    r.w = m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15] * v.w;
    return r;
}
// affine 3x4 row-major; the association order is part of the parity contract (DESIGN.md "instance transforms")
SKH_DI v3 xform_point(const float* m, const v3& p)
{
    return mk3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
               ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
// world -> object for POINTS: o' = R^-1 (p - T), R^-1 = the 3x3 of the instance's world-to-object record, T = the translation of
// its object-to-world transform, kept in the record's fourth column (m[3], m[7], m[11]).  R^-1 p + t' with t' = -R^-1 T -- the
// usual 3x4 inverse, OptiX's too -- cancels two large terms when the instance sits far from the origin: the rounding of p is
// amplified by |R^-1| and a squashed instance a few float spacings thick gets an object-space origin that is noise.  Subtracting
// first is exact for p near T (Sterbenz) and moves that limit out by orders of magnitude.
#ifndef SKH_ENTRY_REL
#define SKH_ENTRY_REL 1 // (0: the R^-1 p + t' form, kept for A/B timing only -- the parity tests need 1)
#endif
SKH_DI v3 xform_point_rel(const float* m, const v3& p)
{
#if !SKH_ENTRY_REL
    return mk3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
               ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
#endif
    const float x = p.x - m[3], y = p.y - m[7], z = p.z - m[11];
    return mk3((m[0] * x + m[1] * y) + m[2] * z, (m[4] * x + m[5] * y) + m[6] * z, (m[8] * x + m[9] * y) + m[10] * z);
}
SKH_DI v3 xform_vector(const float* m, const v3& v)
{
    return mk3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
               (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}
SKH_DI v3 xform_normal(const float* w2o, const v3& n) // transpose(w2o 3x3) * n
{
    return mk3((w2o[0] * n.x + w2o[4] * n.y) + w2o[8] * n.z, (w2o[1] * n.x + w2o[5] * n.y) + w2o[9] * n.z,
               (w2o[2] * n.x + w2o[6] * n.y) + w2o[10] * n.z);
}

// =================================================================================================
// Sampler: src/render/optix/RandomSampler.h.  Owen-scrambled Sobol, 5 dimensions, Morton-indexed.
// =================================================================================================
#define SKH_ONE_MINUS_EPS 0x1.fffffep-1f // RandomSampler.h:6

enum
{
    DIM_PIXEL_X = 0,
    DIM_PIXEL_Y,
    DIM_LIGHT_ID,
    DIM_LIGHT_X,
    DIM_LIGHT_Y,
    DIM_BSDF0,
    DIM_BSDF1,
    DIM_BSDF2,
    DIM_BSDF3,
    DIM_RR,
    DIM_COUNT
}; // RandomSampler.h:13-26

// Sobol generator matrices (RandomSampler.h:139-164): filled by the host at context creation from the
// Joe-Kuo recurrences (skh_capi.h: init_sobol_table) and checked against the reference table by the tests.
__constant__ uint32_t c_sobol[5][32];
// the same matrices folded per index byte: g_sobol_lut[(dim * 4 + byte) * 256 + v] = XOR of the columns selected by v.
// k_shade stages it in LDS (20 KB): sobol_uint becomes 4 LDS reads + 3 XORs instead of 32 x (bfe, and, xor).
#define SKH_SOBOL_LUT_WORDS (5 * 4 * 256)
__device__ uint32_t g_sobol_lut[SKH_SOBOL_LUT_WORDS];

SKH_DI uint32_t hash_murmur(uint32_t x) // RandomSampler.h:86-95
{
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
SKH_DI uint32_t hash_combine(uint32_t seed, uint32_t v) // RandomSampler.h:50-53
{
    return seed ^ (v + (seed << 6) + (seed >> 2));
}
SKH_DI uint32_t part1by1(uint32_t x) // RandomSampler.h:115-123
{
    x &= 0x0000ffffu;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    x = (x ^ (x << 1)) & 0x55555555u;
    return x;
}
SKH_DI uint32_t compact1by1(uint32_t x)
{
    x &= 0x55555555u;
    x = (x ^ (x >> 1)) & 0x33333333u;
    x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu;
    x = (x ^ (x >> 8)) & 0x0000ffffu;
    return x;
}
SKH_DI uint32_t encode_morton2(uint32_t x, uint32_t y) // RandomSampler.h:125-128
{
    return (part1by1(y) << 1) + part1by1(x);
}
SKH_DI uint32_t sobol_uint(uint32_t index, uint32_t dim) // RandomSampler.h:166-175 (XOR of selected columns)
{
    uint32_t X = 0;
#pragma unroll
    for (int bit = 0; bit < 32; ++bit)
        X ^= (0u - ((index >> bit) & 1u)) & c_sobol[dim][bit];
    return X;
}
SKH_DI uint32_t laine_karras_permutation(uint32_t value, uint32_t seed) // RandomSampler.h:182-190
{
    value += seed;
    value ^= value * 0x6c50b47cu;
    value ^= value * 0xb82f1e52u;
    value ^= value * 0xc7afe638u;
    value ^= value * 0x8d22f6e6u;
    return value;
}
SKH_DI uint32_t nested_uniform_scramble(uint32_t value, uint32_t seed) // RandomSampler.h:205-211
{
    value = __brev(value);
    value = laine_karras_permutation(value, seed);
    return __brev(value);
}
struct Sampler // SamplerState, RandomSampler.h:28-33
{
    uint32_t seed, sampleIdx, depth;
};
SKH_DI Sampler init_sampler(uint32_t px, uint32_t py, uint32_t pixelSampleIndex, uint32_t maxSampleCount,
                            uint32_t seed) // RandomSampler.h:130-137
{
    Sampler s;
    s.seed = seed;
    s.sampleIdx = encode_morton2(px, py) * maxSampleCount + pixelSampleIndex;
    s.depth = 0;
    return s;
}
static_assert(DIM_COUNT % 5 == 0, "(dim + depth * DIM_COUNT) % 5 == dim % 5 needs DIM_COUNT to be a multiple of 5");
// random<Dim>(state): RandomSampler.h:213-226, including the (Dim + depth*10) % 5 aliasing of dimensions
// ((dim + depth * 10) % 5 == dim % 5: written that way so that aliased dimensions are visibly the same value)
SKH_DI float sampler_random(const Sampler& s, uint32_t dim)
{
    const uint32_t dimension = dim % 5u;
    uint32_t seed = hash_murmur(s.seed + s.depth);
    const uint32_t index = nested_uniform_scramble(s.sampleIdx, seed);
    const uint32_t r = nested_uniform_scramble(sobol_uint(index, dimension), hash_combine(seed, dimension));
    return fminf((float)r * 0x1p-32f, SKH_ONE_MINUS_EPS);
}

// same value as sampler_random, Sobol point from the byte-folded table (lut = LDS copy of g_sobol_lut)
SKH_DI float sampler_random_lut(const Sampler& s, uint32_t dim, const uint32_t* lut)
{
    const uint32_t dimension = dim % 5u;
    uint32_t seed = hash_murmur(s.seed + s.depth);
    const uint32_t index = nested_uniform_scramble(s.sampleIdx, seed);
    const uint32_t* t = lut + dimension * 1024u;
    const uint32_t X = t[index & 255u] ^ t[256u + ((index >> 8) & 255u)] ^ t[512u + ((index >> 16) & 255u)] ^ t[768u + (index >> 24)];
    const uint32_t r = nested_uniform_scramble(X, hash_combine(seed, dimension));
    return fminf((float)r * 0x1p-32f, SKH_ONE_MINUS_EPS);
}

// =================================================================================================
// Camera ray: src/render/optix/OptixRender.cu:38-58
// =================================================================================================
SKH_DI void generate_camera_ray(uint32_t px, uint32_t py, uint32_t width, uint32_t height, const float* clipToView,
                                const float* viewToWorld, float jx, float jy, v3& origin, v3& direction)
{
    const float posx = (float)px + jx;
    const float posy = (float)py + jy;
    const float ndcx = (posx / (float)width) * 2.0f - 1.0f;
    const float ndcy = (posy / (float)height) * 2.0f - 1.0f;
    const v4 viewSpace = mul44(clipToView, mk4(ndcx, ndcy, 1.0f, 1.0f));
    const v4 wdir = mul44(viewToWorld, mk4(viewSpace.x, viewSpace.y, viewSpace.z, 0.0f));
    origin = mk3(mul44(viewToWorld, mk4(0.0f, 0.0f, 0.0f, 1.0f)));
    direction = normalize(mk3(wdir));
}

// =================================================================================================
// Lights: include/render/Lights.h
// =================================================================================================
#define SKH_PI 3.14159265358979323846f // M_PIf

struct Light // UniformLight, Lights.h:5-14 (112 B)
{
    v4 points[4];
    v4 color;
    v4 normal;
    int32_t type;
    float halfAngle;
    float pad0, pad1;
};
struct LightSample // LightSampleData, Lights.h:16-26
{
    v3 pointOnLight;
    float pdf;
    v3 normal;
    float area;
    v3 L;
    float distToLight;
};
SKH_DI float mis_weight_balance(float a, float b) // Lights.h:28-31
{
    return 1.0f / (1.0f + (b / a));
}
SKH_DI float calc_light_area(const Light& l) // Lights.h:33-52
{
    float area = 0.0f;
    if (l.type == 0)
    {
        const v3 e1 = mk3(l.points[1]) - mk3(l.points[0]);
        const v3 e2 = mk3(l.points[3]) - mk3(l.points[0]);
        area = length(cross(e1, e2));
    }
    else if (l.type == 1)
        area = SKH_PI * l.points[0].x * l.points[0].x;
    else if (l.type == 2)
        area = 4.0f * SKH_PI * l.points[0].x * l.points[0].x;
    return area;
}
SKH_DI v3 calc_light_normal(const Light& l, const v3& hitPoint) // Lights.h:54-74
{
    v3 norm = mk3(0.0f);
    if (l.type == 0)
    {
        const v3 e1 = mk3(l.points[1]) - mk3(l.points[0]);
        const v3 e2 = mk3(l.points[3]) - mk3(l.points[0]);
        norm = -normalize(cross(e1, e2));
    }
    else if (l.type == 1)
        norm = mk3(l.normal);
    else if (l.type == 2)
        norm = normalize(hitPoint - mk3(l.points[1]));
    return norm;
}
SKH_DI void fill_light_data(const Light& l, const v3& hitPoint, LightSample& d) // Lights.h:76-84
{
    d.area = calc_light_area(l);
    d.normal = calc_light_normal(l, hitPoint);
    const v3 toLight = d.pointOnLight - hitPoint;
    const float lenToLight = length(toLight);
    d.L = toLight / lenToLight;
    d.distToLight = lenToLight;
}
struct SphQuad // Lights.h:86-94
{
    v3 o, x, y, z;
    float z0, z0sq, x0, y0, y0sq, x1, y1, y1sq, b0, b1, b0sq, k, S;
};
SKH_DI SphQuad sph_quad_init(const Light& l, const v3& o) // Lights.h:97-153
{
    SphQuad q;
    const v3 ex = mk3(l.points[1]) - mk3(l.points[0]);
    const v3 ey = mk3(l.points[3]) - mk3(l.points[0]);
    const v3 s = mk3(l.points[0]);
    const float exl = length(ex);
    const float eyl = length(ey);
    q.o = o;
    q.x = ex / exl;
    q.y = ey / eyl;
    q.z = cross(q.x, q.y);
    const v3 d = s - o;
    q.z0 = dot(d, q.z);
    if (q.z0 > 0)
    {
        q.z = q.z * -1.0f;
        q.z0 *= -1.0f;
    }
    q.z0sq = q.z0 * q.z0;
    q.x0 = dot(d, q.x);
    q.y0 = dot(d, q.y);
    q.x1 = q.x0 + exl;
    q.y1 = q.y0 + eyl;
    q.y0sq = q.y0 * q.y0;
    q.y1sq = q.y1 * q.y1;
    const v3 v00 = mk3(q.x0, q.y0, q.z0), v01 = mk3(q.x0, q.y1, q.z0), v10 = mk3(q.x1, q.y0, q.z0),
             v11 = mk3(q.x1, q.y1, q.z0);
    const v3 n0 = normalize(cross(v00, v10));
    const v3 n1 = normalize(cross(v10, v11));
    const v3 n2 = normalize(cross(v11, v01));
    const v3 n3 = normalize(cross(v01, v00));
    const float g0 = skm::acosf_(-dot(n0, n1));
    const float g1 = skm::acosf_(-dot(n1, n2));
    const float g2 = skm::acosf_(-dot(n2, n3));
    const float g3 = skm::acosf_(-dot(n3, n0));
    q.b0 = n0.z;
    q.b1 = n2.z;
    q.b0sq = q.b0 * q.b0;
    q.k = 2.0f * SKH_PI - g2 - g3;
    q.S = g0 + g1 - q.k;
    return q;
}
SKH_DI v3 sph_quad_sample(const SphQuad& q, float u, float v) // Lights.h:155-189
{
    const float au = u * q.S + q.k;
    const float fu = (skm::cosf_(au) * q.b0 - q.b1) / skm::sinf_(au);
    float cu = 1.0f / sqrtf(fu * fu + q.b0sq) * (fu > 0.0f ? 1.0f : -1.0f);
    cu = clampf(cu, -1.0f, 1.0f);
    float xu = -(cu * q.z0) / sqrtf(1.0f - cu * cu);
    xu = clampf(xu, q.x0, q.x1);
    const float d = sqrtf(xu * xu + q.z0sq);
    const float h0 = q.y0 / sqrtf(d * d + q.y0sq);
    const float h1 = q.y1 / sqrtf(d * d + q.y1sq);
    const float hv = h0 + v * (h1 - h0);
    const float hv2 = hv * hv;
    const float eps = 1e-5f;
    const float yv = (hv < 1.0f - eps) ? (hv * d) / sqrtf(1 - hv2) : q.y1;
    return (q.o + xu * q.x + yv * q.y + q.z0 * q.z);
}
SKH_DI float get_light_pdf(const Light& l, const v3& lightHitPoint, const v3& surfaceHitPoint) // Lights.h:201-243
{
    switch (l.type)
    {
    case 0: {
        LightSample d;
        d.pointOnLight = lightHitPoint;
        fill_light_data(l, surfaceHitPoint, d);
        return d.distToLight * d.distToLight / (dot(-d.L, d.normal) * d.area);
    }
    case 2:
        return 1.0f / (4.0f * SKH_PI);
    case 3:
        return 1.0f / (2.0f * SKH_PI * (1.0f - skm::cosf_(l.halfAngle)));
    default:
        break;
    }
    return 0.0f;
}
SKH_DI LightSample sample_rect_light_uniform(const Light& l, float ux, float uy, const v3& hitPoint) // Lights.h:277-289
{
    LightSample d;
    const v3 e1 = mk3(l.points[1]) - mk3(l.points[0]);
    const v3 e2 = mk3(l.points[3]) - mk3(l.points[0]);
    d.pointOnLight = mk3(l.points[0]) + e1 * ux + e2 * uy;
    fill_light_data(l, hitPoint, d);
    d.pdf = d.distToLight * d.distToLight / (-dot(d.L, d.normal) * d.area);
    return d;
}
SKH_DI LightSample sample_rect_light(const Light& l, float ux, float uy, const v3& hitPoint) // Lights.h:245-275
{
    LightSample d;
    const v3 e1 = mk3(l.points[1]) - mk3(l.points[0]);
    const v3 e2 = mk3(l.points[3]) - mk3(l.points[0]);
    const SphQuad quad = sph_quad_init(l, hitPoint);
    if (quad.S <= 0.0f)
    {
        d.pdf = 0.0f;
        d.pointOnLight = mk3(l.points[0]) + e1 * ux + e2 * uy;
        fill_light_data(l, hitPoint, d);
        return d;
    }
    if (quad.S < 1e-3f)
    {
        d.pointOnLight = mk3(l.points[0]) + e1 * ux + e2 * uy;
        fill_light_data(l, hitPoint, d);
        d.pdf = d.distToLight * d.distToLight / (-dot(d.L, d.normal) * d.area);
        return d;
    }
    d.pointOnLight = sph_quad_sample(quad, ux, uy);
    fill_light_data(l, hitPoint, d);
    d.pdf = 1.0f / quad.S;
    return d;
}
SKH_DI void create_coordinate_system(const v3& N, v3& Nt, v3& Nb) // Lights.h:291-300
{
    if (fabsf(N.x) > fabsf(N.y))
    {
        const float invLen = 1.0f / sqrtf(N.x * N.x + N.z * N.z);
        Nt = mk3(-N.z * invLen, 0.0f, N.x * invLen);
    }
    else
    {
        const float invLen = 1.0f / sqrtf(N.y * N.y + N.z * N.z);
        Nt = mk3(0.0f, N.z * invLen, -N.y * invLen);
    }
    Nb = cross(N, Nt);
}
// Lights.h:302-317: the double literals in the reference promote parts of this function to fp64
SKH_DI v3 sample_cone(float ux, float uy, float angle, const v3& direction, float& pdf)
{
    const float phi = (float)(2.0 * (double)SKH_PI * (double)ux);
    const float cosTheta = (float)(1.0 - (double)uy * (1.0 - (double)skm::cosf_(angle)));
    const float sinTheta = (float)sqrt(1.0 - (double)(cosTheta * cosTheta));
    v3 u, v;
    create_coordinate_system(direction, u, v);
    const v3 sampledDir = normalize(skm::cosf_(phi) * sinTheta * u + skm::sinf_(phi) * sinTheta * v + cosTheta * direction);
    pdf = (float)(1.0 / (2.0 * (double)SKH_PI * (1.0 - (double)skm::cosf_(angle))));
    return sampledDir;
}
SKH_DI LightSample sample_distant_light(const Light& l, float ux, float uy) // Lights.h:319-333
{
    LightSample d;
    float pdf = 0.0f;
    const v3 coneSample = sample_cone(ux, uy, l.halfAngle, -mk3(l.normal), pdf);
    d.area = 0.0f;
    d.distToLight = 1e9f;
    d.L = coneSample;
    d.normal = mk3(l.normal);
    d.pdf = pdf;
    d.pointOnLight = coneSample;
    return d;
}
SKH_DI LightSample sample_sphere_light(const Light& l, float ux, float uy, const v3& hitPoint) // Lights.h:335-362
{
    LightSample d;
    const float cosTheta = 1.0f - 2.0f * ux;
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    const float phi = 2.0f * SKH_PI * uy;
    const float radius = l.points[0].x;
    const v3 sphereDirection = mk3(sinTheta * skm::cosf_(phi), sinTheta * skm::sinf_(phi), cosTheta);
    const v3 lightPoint = mk3(l.points[1]) + radius * sphereDirection;
    d.L = normalize(lightPoint - hitPoint);
    d.distToLight = length(lightPoint - hitPoint);
    d.area = 0.0f;
    d.normal = sphereDirection;
    d.pdf = 1.0f / (4.0f * SKH_PI);
    d.pointOnLight = lightPoint;
    return d;
}

// =================================================================================================
// Hit reconstruction helpers: src/render/optix/OptixRender_radiance_closest_hit.cu:199-254
// =================================================================================================
SKH_DI v3 unpack_normal(uint32_t val) // closest_hit.cu:236-244
{
    v3 n;
    n.z = (float)((val & 0xfff00000u) >> 20) / 511.99999f * 2.0f - 1.0f;
    n.y = (float)((val & 0x000ffc00u) >> 10) / 511.99999f * 2.0f - 1.0f;
    n.x = (float)(val & 0x000003ffu) / 511.99999f * 2.0f - 1.0f;
    return n;
}
SKH_DI void unpack_uv(uint32_t val, float& u, float& v) // closest_hit.cu:247-254
{
    v = (float)((val & 0xffff0000u) >> 16) / 16383.99999f * 20.0f - 10.0f;
    u = (float)(val & 0x0000ffffu) / 16383.99999f * 20.0f - 10.0f;
}
// 2-D texture lookup = tex2D<float4> of the reference's texture objects (uchar4, normalized float, linear filter, wrap,
// normalized coordinates: OptixRender.cpp:1191-1264, texture_support_cuda.h:287-313), restated from the CUDA programming
// guide's linear-filtering formula with 1.8 fixed-point weights.
SKH_DI void tex_axis(float u, uint32_t n, uint32_t& i0, uint32_t& i1, float& a)
{
    const float x = (u - floorf(u)) * (float)n - 0.5f;
    const float fl = floorf(x);
    a = floorf((x - fl) * 256.0f + 0.5f) * (1.0f / 256.0f);
    int i = (int)fl;
    i = i < 0 ? i + (int)n : i;
    i0 = (uint32_t)i % n;
    i1 = (i0 + 1u) % n;
}
SKH_DI v4 texel_rgba8(uint32_t t)
{
    return mk4((float)(t & 0xffu) / 255.0f, (float)((t >> 8) & 0xffu) / 255.0f, (float)((t >> 16) & 0xffu) / 255.0f, (float)(t >> 24) / 255.0f);
}
SKH_DI v4 tex_lookup_rgba8(const uint32_t* __restrict__ texels, const uint4 desc /*offset, width, height*/, float u, float v)
{
    uint32_t x0, x1, y0, y1;
    float a, b;
    tex_axis(u, desc.y, x0, x1, a);
    tex_axis(v, desc.z, y0, y1, b);
    const uint32_t* base = texels + desc.x;
    const v4 t00 = texel_rgba8(base[y0 * desc.y + x0]), t10 = texel_rgba8(base[y0 * desc.y + x1]);
    const v4 t01 = texel_rgba8(base[y1 * desc.y + x0]), t11 = texel_rgba8(base[y1 * desc.y + x1]);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return mk4(((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x, ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y,
               ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z, ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w);
}
SKH_DI v3 interpolate_attrib(const v3& a1, const v3& a2, const v3& a3, float bx, float by) // closest_hit.cu:199-205
{
    return a1 * (1.0f - bx - by) + a2 * bx + a3 * by;
}
SKH_DI v3 offset_ray(const v3& p, const v3& n) // closest_hit.cu:218-233
{
    const float origin = 1.0f / 32.0f;
    const float float_scale = 1.0f / 65536.0f;
    const float int_scale = 256.0f;
    const int ofx = (int)(int_scale * n.x), ofy = (int)(int_scale * n.y), ofz = (int)(int_scale * n.z);
    const v3 p_i = mk3(__int_as_float(__float_as_int(p.x) + ((p.x < 0) ? -ofx : ofx)),
                       __int_as_float(__float_as_int(p.y) + ((p.y < 0) ? -ofy : ofy)),
                       __int_as_float(__float_as_int(p.z) + ((p.z < 0) ? -ofz : ofz)));
    return mk3(fabsf(p.x) < origin ? p.x + float_scale * n.x : p_i.x, fabsf(p.y) < origin ? p.y + float_scale * n.y : p_i.y,
               fabsf(p.z) < origin ? p.z + float_scale * n.z : p_i.z);
}

// Cubic B-spline segment as a polynomial: cuda/curve.h:177-187, 237-240, 252-260, 272-275
struct CubicPoly
{
    v4 p[4];
};
SKH_DI void cubic_from_bspline(CubicPoly& c, const v4* q) // curve.h:177-187
{
    c.p[0] = (q[0] * (-1.0f) + q[1] * (3.0f) + q[2] * (-3.0f) + q[3]) / 6.0f;
    c.p[1] = (q[0] * (3.0f) + q[1] * (-6.0f) + q[2] * (3.0f)) / 6.0f;
    c.p[2] = (q[0] * (-3.0f) + q[2] * (3.0f)) / 6.0f;
    c.p[3] = (q[0] * (1.0f) + q[1] * (4.0f) + q[2] * (1.0f)) / 6.0f;
}
SKH_DI v4 cubic_position(const CubicPoly& c, float u) // curve.h:237-240
{
    return (((c.p[0] * u) + c.p[1]) * u + c.p[2]) * u + c.p[3];
}
SKH_DI v4 cubic_velocity(const CubicPoly& c, float u) // curve.h:252-260
{
    if (u == 0)
        u = 0.000001f;
    if (u == 1)
        u = 0.999999f;
    return ((3.0f * c.p[0] * u) + 2.0f * c.p[1]) * u + c.p[2];
}
SKH_DI v4 cubic_acceleration(const CubicPoly& c, float u) // curve.h:272-275
{
    return 6.0f * c.p[0] * u + 2.0f * c.p[1];
}
SKH_DI v3 curve_surface_normal(const CubicPoly& bc, float u, v3& ps) // surfaceNormal<Cubic,2>: curve.h:306-353
{
    v3 normal;
    if (u == 0.0f)
        normal = -mk3(cubic_velocity(bc, 0));
    else if (u == 1.0f)
        normal = mk3(cubic_velocity(bc, 1));
    else
    {
        const v4 p4 = cubic_position(bc, u);
        const v3 p = mk3(p4);
        const float r = p4.w;
        const v4 d4 = cubic_velocity(bc, u);
        const v3 d = mk3(d4);
        const float dr = d4.w;
        float dd = dot(d, d);
        v3 o1 = ps - p;
        o1 = o1 - (dot(o1, d) / dd) * d;
        o1 = o1 * (r / length(o1));
        ps = p + o1;
        dd -= dot(mk3(cubic_acceleration(bc, u)), o1);
        normal = dd * o1 - (dr * r) * d;
    }
    return normalize(normal);
}

// =================================================================================================
// Accumulation: postprocessing/Utils.h:5-14, OptixRender.cu:60-78
// =================================================================================================
SKH_DI v3 tonemap(v3 color, const v3& exposure)
{
    color = color * exposure;
    return color / (color + mk3(1.0f));
}
SKH_DI v3 inverse_tonemap(const v3& color, const v3& exposure)
{
    return color / (exposure - color * exposure);
}
SKH_DI v3 accumulate(const v3& prev, const v3& value, const v3& exposure, uint32_t subFrameIndex)
{
    v3 accumColor = value;
    if (subFrameIndex > 0)
    {
        const float a = 1.0f / (float)(subFrameIndex + 1);
        accumColor = inverse_tonemap(lerp3(tonemap(prev, exposure), tonemap(accumColor, exposure), a), exposure);
    }
    return accumColor;
}

// =================================================================================================
// Ray / primitive intersection (own definition; the reference uses closed OptiX).  See DESIGN.md.
// =================================================================================================
struct RayShear
{
    int perm; // dominant axis kz in bits 0..1; bit 2: kx and ky swapped (d[kz] < 0)
    float Sx, Sy, Sz;
};
// (v[kx], v[ky], v[kz]) with kx = kz + 1, ky = kz + 2 (mod 3), swapped when d[kz] < 0.  Written as selects on locals so
// that it compiles to v_cndmask (a chain of compares on a `const v3&` became a branchy switch: ~25 instructions and
// four branches per component, nine components per triangle).
SKH_DI v3 shear_permute(const v3& vv, int perm)
{
    const float x = vv.x, y = vv.y, z = vv.z;
    const int kz = perm & 3;
    const bool z0 = kz == 0, z1 = kz == 1, fl = perm >= 4;
    const float a = z0 ? y : (z1 ? z : x);
    const float b = z0 ? z : (z1 ? x : y);
    const float c = z0 ? x : (z1 ? y : z);
    return mk3(fl ? b : a, fl ? a : b, c);
}
SKH_DI RayShear make_shear(const v3& d)
{
    RayShear s;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    const float dkz = kz == 0 ? d.x : (kz == 1 ? d.y : d.z);
    s.perm = kz | (dkz < 0.0f ? 4 : 0);
    const v3 dp = shear_permute(d, s.perm);
    s.Sz = 1.0f / dp.z; // one division; the shear factors use the reciprocal (same definition in the oracle)
    s.Sx = dp.x * s.Sz;
    s.Sy = dp.y * s.Sz;
    return s;
}
// watertight edge-function test (Woop, Benthin, Wald 2013); accepts tmin < t <= tmax
SKH_DI bool intersect_triangle(const v3& o, const RayShear& s, float tmin, float tmax, const v3& p0, const v3& p1,
                               const v3& p2, float& t_out, float& u_out, float& v_out, uint32_t* prof = nullptr /*lane-profile build*/)
{
    const v3 A = shear_permute(p0 - o, s.perm), B = shear_permute(p1 - o, s.perm), C = shear_permute(p2 - o, s.perm);
    const float Akz = A.z, Bkz = B.z, Ckz = C.z;
    const float Ax = A.x - s.Sx * Akz;
    const float Ay = A.y - s.Sy * Akz;
    const float Bx = B.x - s.Sx * Bkz;
    const float By = B.y - s.Sy * Bkz;
    const float Cx = C.x - s.Sx * Ckz;
    const float Cy = C.y - s.Sy * Ckz;
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
        if (prof)
            *prof |= 1u;
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
    if (prof)
        *prof |= 2u;
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
    if (prof)
        *prof |= 4u;
    const float rcpDet = 1.0f / det;
    const float t = T * rcpDet;
    if (!(t > tmin && t <= tmax))
        return false;
    t_out = t;
    u_out = V * rcpDet;
    v_out = W * rcpDet;
    return true;
}
SKH_DI void onb_from_z(const v3& n, v3& b1, v3& b2) // Duff et al. 2017
{
    const float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    b1 = mk3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    b2 = mk3(b, sign + n.y * n.y * a, -n.y);
}
// Phantom ray-hair intersector (Reshetov & Luebke 2018), round cubic B-spline, varying radius, no end caps.
SKH_DI bool intersect_curve_segment(const v3& o, const v3& d, float tmin, float tmax, const v4* q, float& t_out, float& u_out)
{
    const float dlen = sqrtf(dot(d, d));
    const float inv_dlen = 1.0f / dlen;
    const v3 dn = d * inv_dlen;
    v3 bx, by;
    onb_from_z(dn, bx, by);
    v4 qc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const v3 p = mk3(q[i]) - o;
        qc[i] = mk4(dot(p, bx), dot(p, by), dot(p, dn), q[i].w);
    }
    CubicPoly poly;
    cubic_from_bspline(poly, qc);
    const v4 e0 = cubic_position(poly, 0.0f);
    const v4 e1 = cubic_position(poly, 1.0f);
    float tstart = (e1.z - e0.z) > 0.0f ? 0.0f : 1.0f;
    // both ends are always tried and the nearer accepted root wins: the roots do not depend on [tmin, tmax], so the closest
    // hit does not depend on the order in which segments are visited (returning the first accepted root did)
    bool found = false;
    for (int ep = 0; ep < 2; ++ep)
    {
        float t = tstart;
        float told = 0.0f, dt1 = 0.0f, dt2 = 0.0f;
        for (int i = 0; i < 40; ++i)
        {
            const v4 c4 = cubic_position(poly, t);
            const v4 d4 = ((3.0f * poly.p[0] * t) + 2.0f * poly.p[1]) * t + poly.p[2];
            const v3 c0 = mk3(c4), cd = mk3(d4);
            const float r = c4.w, dr = d4.w;
            // ray / tangent-cone intersection
            const float r2 = r * r;
            const float drr = r * dr;
            float ddd = cd.x * cd.x + cd.y * cd.y;
            float dp = c0.x * c0.x + c0.y * c0.y;
            const float cdd = c0.x * cd.x + c0.y * cd.y;
            const float cxd = c0.x * cd.y - c0.y * cd.x;
            const float c = ddd;
            const float b = cd.z * (drr - cdd);
            const float cdz2 = cd.z * cd.z;
            ddd += cdz2;
            const float a = ((2.0f * drr * cdd + cxd * cxd) - ddd * r2) + dp * cdz2;
            const float det = b * b - a * c;
            const float s = (b - (det > 0.0f ? sqrtf(det) : 0.0f)) / c;
            float dt = (s * cd.z - cdd) / ddd;
            const bool phantom = !(det > 0.0f);
            if (!phantom && fabsf(dt) < 5e-5f)
            {
                const float sw = (s + c0.z) * inv_dlen;
                // A converged point lies ON the tube: its distance from the curve point, less the part along the tangent (dt |c'|), is the radius there.  A ray (nearly)
                // parallel to the tangent makes the cone's quadratic degenerate -- c = |c'_xy|^2 -> 0, b - sqrt(det) cancels to 0, dt comes out small and det > 0 by
                // rounding -- and the iteration "converges" at once on a point far off the tube (round 6, fuzz_render seed 5483: 0.49 from a curve point of radius
                // 0.096; found or not depending on which boxes the ray met).  Such a root is not a hit: every bound the hierarchies keep (hull + largest radius) relies on that -- the radius along the
                // segment stays inside the hull of the control radii, and an accepted point is within 1.0005 of the cone's radius of the curve point.
                const float radial2 = (s * s + dp) - (dt * dt) * ddd;
                const float rb = r + dr * dt; // the cone's radius at that offset along the tangent (linear: exact for the cone)
                if (radial2 <= 1.001f * (rb * rb) && sw > tmin && sw <= tmax && t >= 0.0f && t <= 1.0f && (!found || sw < t_out))
                {
                    t_out = sw;
                    u_out = t;
                    found = true;
                }
                break;
            }
            if (phantom && fabsf(dt) < 5e-5f)
                break; // converged onto a point the ray does not touch (the closest approach of a miss): same rule as the oracle
            dt = fminf(dt, 0.5f);
            dt = fmaxf(dt, -0.5f);
            dt1 = dt2;
            dt2 = dt;
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
                t += dt;
            }
            if (!(t >= 0.0f && t <= 1.0f))
                break;
        }
        tstart = 1.0f - tstart;
    }
    return found;
}

// =================================================================================================
// "MDL-equivalent" BSDF set behind the mdlcode_init/sample/evaluate protocol (closest_hit.cu:477-605).
// Definitions: DESIGN.md "BSDF set".  Lambert semantics: metal/shaders/pathtrace.metal:164-201.
// =================================================================================================
enum
{
    EV_ABSORB = 0,
    EV_DIFFUSE = 1,
    EV_GLOSSY = 2,
    EV_SPECULAR = 4,
    EV_REFLECTION = 8,
    EV_TRANSMISSION = 16
};
struct Material // skh_material, 64 B
{
    uint32_t type;
    float base_color[3];
    float roughness, metallic, specular, ior;
    uint32_t base_color_texture, normal_texture; // 1-based texture ids, 0 = none
    float reserved[6];
};
struct BsdfSample
{
    v3 k2, bsdf_over_pdf;
    float pdf;
    int event_type;
};
struct BsdfEval
{
    v3 bsdf_diffuse, bsdf_glossy;
    float pdf;
};
SKH_DI v3 cosine_hemisphere(float u1, float u2, float& cosTheta)
{
    const float r = sqrtf(u1);
    const float phi = 2.0f * SKH_PI * u2;
    cosTheta = sqrtf(fmaxf(0.0f, 1.0f - u1));
    return mk3(r * skm::cosf_(phi), r * skm::sinf_(phi), cosTheta);
}
SKH_DI v3 schlick3(const v3& f0, float c)
{
    const float m = clampf(1.0f - c, 0.0f, 1.0f);
    const float m2 = m * m;
    const float m5 = m2 * m2 * m;
    return f0 + (mk3(1.0f) - f0) * m5;
}
SKH_DI float ggx_D(float alpha, float nh)
{
    const float a2 = alpha * alpha;
    const float d = nh * nh * (a2 - 1.0f) + 1.0f;
    return a2 / (SKH_PI * d * d);
}
SKH_DI float ggx_lambda(float alpha, float cosT)
{
    const float c2 = cosT * cosT;
    const float t2 = fmaxf(0.0f, 1.0f - c2) / fmaxf(c2, 1e-20f);
    return 0.5f * (sqrtf(1.0f + alpha * alpha * t2) - 1.0f);
}
SKH_DI v3 ggx_sample_vndf(const v3& Ve, float alpha, float u1, float u2) // Heitz 2018
{
    const v3 Vh = normalize(mk3(alpha * Ve.x, alpha * Ve.y, Ve.z));
    const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    const v3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : mk3(1.0f, 0.0f, 0.0f);
    const v3 T2 = cross(Vh, T1);
    const float r = sqrtf(u1);
    const float phi = 2.0f * SKH_PI * u2;
    const float t1 = r * skm::cosf_(phi);
    float t2 = r * skm::sinf_(phi);
    const float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1)) + s * t2;
    const v3 Nh = t1 * T1 + t2 * T2 + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(mk3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)));
}
struct PbrTerms
{
    v3 diffuse_albedo, f0;
    float alpha, p_spec;
};
SKH_DI PbrTerms pbr_terms(const Material& m)
{
    PbrTerms t;
    const v3 base = mk3(m.base_color[0], m.base_color[1], m.base_color[2]);
    const float metallic = clampf(m.metallic, 0.0f, 1.0f);
    t.diffuse_albedo = base * (1.0f - metallic);
    const float d = 0.08f * m.specular;
    t.f0 = mk3(d) + (base - mk3(d)) * metallic;
    const float r = fmaxf(m.roughness, 0.05f);
    t.alpha = r * r;
    t.p_spec = 0.5f + 0.5f * metallic;
    return t;
}
SKH_DI void pbr_eval_local(const PbrTerms& t, const v3& wo, const v3& wi, v3& fd, v3& fs, float& pdf)
{
    fd = mk3(0.0f);
    fs = mk3(0.0f);
    pdf = 0.0f;
    if (wo.z <= 0.0f || wi.z <= 0.0f)
        return;
    const v3 h = normalize(wo + wi);
    const float oh = fmaxf(dot(wo, h), 0.0f);
    const v3 F = schlick3(t.f0, oh);
    const float D = ggx_D(t.alpha, h.z);
    const float lo = ggx_lambda(t.alpha, wo.z), li = ggx_lambda(t.alpha, wi.z);
    const float G2 = 1.0f / (1.0f + lo + li);
    const float G1 = 1.0f / (1.0f + lo);
    fs = F * (D * G2 / (4.0f * wo.z));
    const v3 Fo = schlick3(t.f0, wo.z);
    fd = t.diffuse_albedo * (mk3(1.0f) - Fo) * (wi.z / SKH_PI);
    const float pdf_s = G1 * D / (4.0f * wo.z);
    const float pdf_d = wi.z / SKH_PI;
    pdf = t.p_spec * pdf_s + (1.0f - t.p_spec) * pdf_d;
}
SKH_DI float fresnel_dielectric(float cosi, float eta, float& cost)
{
    const float sin2t = eta * eta * fmaxf(0.0f, 1.0f - cosi * cosi);
    if (sin2t >= 1.0f)
    {
        cost = 0.0f;
        return 1.0f;
    }
    cost = sqrtf(1.0f - sin2t);
    const float rs = (eta * cosi - cost) / (eta * cosi + cost);
    const float rp = (cosi - eta * cost) / (cosi + eta * cost);
    return 0.5f * (rs * rs + rp * rp);
}
// ------------------------------------------------------------------------------------------------------------
// Rough dielectric: OmniGlass with frosting_roughness > 0 (gltfloader.cpp:354-406).  Walter et al. 2007 with the GGX
// distribution, alpha = frosting_roughness^2, visible-normal sampling; same definitions, operation for operation, as
// the CPU checker's twin (rough_glass_*).  Local frame: z = N on k1's side; eta = n1 / n2.
// ------------------------------------------------------------------------------------------------------------
#define SKH_GLASS_SMOOTH_BELOW 1e-3f
SKH_DI void rough_glass_eval_local(float alpha, float eta, const v3& tint, const v3& wo, const v3& wi, v3& f_cos, float& pdf)
{
    f_cos = mk3(0.0f);
    pdf = 0.0f;
    if (wo.z <= 0.0f || wi.z == 0.0f)
        return;
    const bool reflect = wi.z > 0.0f;
    v3 h = reflect ? wo + wi : (wo * eta + wi);
    const float hl = sqrtf(dot(h, h));
    if (!(hl > 0.0f))
        return;
    h = h * (1.0f / hl);
    if (h.z < 0.0f)
        h = -h;
    const float oh = dot(wo, h), ih = dot(wi, h);
    if (oh <= 0.0f || (reflect ? ih <= 0.0f : ih >= 0.0f))
        return;
    float cost;
    const float F = fresnel_dielectric(fminf(oh, 1.0f), eta, cost);
    const float D = ggx_D(alpha, h.z);
    const float lo = ggx_lambda(alpha, wo.z), li = ggx_lambda(alpha, fabsf(wi.z));
    const float G1o = 1.0f / (1.0f + lo);
    const float G2 = 1.0f / (1.0f + lo + li);
    const float pdf_h = G1o * D * oh / wo.z;
    if (reflect)
    {
        const float jac = 1.0f / (4.0f * oh);
        pdf = F * pdf_h * jac;
        f_cos = mk3(F * D * G2 / (4.0f * wo.z));
    }
    else
    {
        const float denom = eta * oh + ih;
        const float jac = fabsf(ih) / (denom * denom);
        pdf = (1.0f - F) * pdf_h * jac;
        f_cos = tint * ((1.0f - F) * D * G2 * oh * jac / wo.z);
    }
}
SKH_DI bool rough_glass_sample_local(float alpha, float eta, const v3& tint, const v3& wo, float u0, float u1, float u2, v3& wi, v3& weight,
                                     float& pdf, bool& transmitted)
{
    const v3 h = ggx_sample_vndf(wo, alpha, u0, u1);
    const float oh = dot(wo, h);
    if (oh <= 0.0f)
        return false;
    float cost;
    const float F = fresnel_dielectric(fminf(oh, 1.0f), eta, cost);
    transmitted = !(u2 < F);
    if (!transmitted)
        wi = h * (2.0f * oh) - wo;
    else
        wi = normalize(h * (eta * oh - cost) - wo * eta);
    if (transmitted ? wi.z >= 0.0f : wi.z <= 0.0f)
        return false;
    v3 f_cos;
    rough_glass_eval_local(alpha, eta, tint, wo, wi, f_cos, pdf);
    if (!(pdf > 0.0f))
        return false;
    weight = f_cos / pdf;
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// Hair: df::chiang_hair_bsdf (the `hair` sub-expression the reference compiles for hair materials, mdlPtxCodeGen.cpp:143-155).
// Chiang et al. 2016 in the pbrt-v3 formulation; parameter layout (include/strelka_hip.h) and every definition as the CPU checker's twin (hair_*), same
// operation order.  h = 2 * text_coords[0].y - 1 with the reference's constant text_coords = 0.5 (closest_hit.cu:445).
// ------------------------------------------------------------------------------------------------------------
#define SKH_HAIR_TEXCOORD_Y 0.5f
SKH_DI float lum3(const v3& c)
{
    return 0.299f * c.x + 0.587f * c.y + 0.114f * c.z;
}
SKH_DI float sqrf_(float x)
{
    return x * x;
}
SKH_DI float safe_sqrtf_(float x)
{
    return sqrtf(fmaxf(0.0f, x));
}
SKH_DI float safe_asinf_(float x)
{
    return skm::asinf_(clampf(x, -1.0f, 1.0f));
}
SKH_DI float hair_pow20(float x)
{
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    return x16 * x4;
}
SKH_DI float hair_pow22(float x)
{
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    return (x16 * x4) * x2;
}
SKH_DI float hair_I0(float x)
{
    const float y = x * x;
    float v = 1.0f / 34519618525593600.0f;
    v = v * y + 1.0f / 106542032486400.0f;
    v = v * y + 1.0f / 416179814400.0f;
    v = v * y + 1.0f / 2123366400.0f;
    v = v * y + 1.0f / 14745600.0f;
    v = v * y + 1.0f / 147456.0f;
    v = v * y + 1.0f / 2304.0f;
    v = v * y + 1.0f / 64.0f;
    v = v * y + 0.25f;
    v = v * y + 1.0f;
    return v;
}
SKH_DI float hair_logI0(float x)
{
    if (x > 12.0f)
        return x + 0.5f * ((-skm::logf_(2.0f * SKH_PI) + skm::logf_(1.0f / x)) + 1.0f / (8.0f * x));
    return skm::logf_(hair_I0(x));
}
// What of M_p / N_p / the sampler depends on the MATERIAL only -- computed once per material (k_hair_consts: the same expressions, operation for operation,
// that the per-call code of rounds 2-5 and the checker evaluate: same bits) instead of per BSDF call: per lobe 1 / v, log(1 / (2 v)), sinh(1 / v) 2 v and
// exp(-2 / v); the trimmed logistic's normalisation and its lower CDF value.  (Round 6: the hair build of k_shade paid 15 % for the shared polynomial
// exp / log, 13 of them per evaluation in terms that never change.)
struct HairConst // 24 floats per material in DevScene::hairConst; lobes R, TT, TRT (the residual lobe uses TRT's)
{
    float v[3], invV[3], log1over2v[3], sinhDen[3], expm2overV[3];
    float s, logisticDen, cdfA;
    float sin2k[3], cos2k[3];
};
SKH_DI float hair_Mp(float cosThetaI, float cosThetaO, float sinThetaI, float sinThetaO, float v, float invV, float log1over2v, float sinhDen)
{
    const float a = cosThetaI * cosThetaO / v;
    const float b = sinThetaI * sinThetaO / v;
    return v <= 0.1f ? skm::expf_((((hair_logI0(a) - b) - invV) + 0.6931f) + log1over2v) : (skm::expf_(-b) * hair_I0(a)) / sinhDen;
}
SKH_DI float hair_logistic(float x, float s)
{
    x = fabsf(x);
    const float e = skm::expf_(-x / s);
    return e / (s * sqrf_(1.0f + e));
}
SKH_DI float hair_logistic_cdf(float x, float s)
{
    return 1.0f / (1.0f + skm::expf_(-x / s));
}
SKH_DI float hair_trimmed_logistic(float x, const HairConst& hc) // over [-pi, pi]
{
    return hair_logistic(x, hc.s) / hc.logisticDen;
}
SKH_DI float hair_sample_trimmed_logistic(float u, const HairConst& hc, float a, float b)
{
    const float x = -hc.s * skm::logf_(1.0f / (u * hc.logisticDen + hc.cdfA) - 1.0f);
    return clampf(x, a, b);
}
SKH_DI float hair_Phi(int p, float gammaO, float gammaT)
{
    return (2.0f * (float)p * gammaT - 2.0f * gammaO) + (float)p * SKH_PI;
}
SKH_DI float hair_Np(float phi, int p, const HairConst& hc, float gammaO, float gammaT)
{
    float dphi = phi - hair_Phi(p, gammaO, gammaT);
    while (dphi > SKH_PI)
        dphi -= 2.0f * SKH_PI;
    while (dphi < -SKH_PI)
        dphi += 2.0f * SKH_PI;
    return hair_trimmed_logistic(dphi, hc);
}
SKH_DI float hair_variance(float roughness)
{
    const float r = fmaxf(roughness, 0.02f);
    return sqrf_((0.726f * r + 0.812f * (r * r)) + 3.7f * hair_pow20(r));
}
SKH_DI void hair_lobe(HairConst& c, int p, float v)
{
    c.v[p] = v;
    c.invV[p] = 1.0f / v;
    c.log1over2v[p] = skm::logf_(1.0f / (2.0f * v));
    c.sinhDen[p] = skm::sinhf_(1.0f / v) * 2.0f * v;
    c.expm2overV[p] = skm::expf_(-2.0f / v);
}
// the material-only part (one thread per material: k_hair_consts; the BSDF probe computes it in place)
SKH_DI HairConst hair_const(const Material& m)
{
    HairConst c;
    const float v0 = hair_variance(m.roughness);
    hair_lobe(c, 0, v0);
    hair_lobe(c, 1, m.metallic > 0.0f ? hair_variance(m.metallic) : 0.25f * v0);
    hair_lobe(c, 2, m.specular > 0.0f ? hair_variance(m.specular) : 4.0f * v0);
    const float bn = fmaxf(m.reserved[3] > 0.0f ? m.reserved[3] : m.roughness, 0.02f);
    c.s = 0.626657069f * ((0.265f * bn + 1.194f * (bn * bn)) + 5.372f * hair_pow22(bn));
    c.logisticDen = hair_logistic_cdf(SKH_PI, c.s) - hair_logistic_cdf(-SKH_PI, c.s);
    c.cdfA = hair_logistic_cdf(-SKH_PI, c.s);
    c.sin2k[0] = skm::sinf_(m.reserved[4]);
    c.cos2k[0] = safe_sqrtf_(1.0f - sqrf_(c.sin2k[0]));
#pragma unroll
    for (int i = 1; i < 3; ++i)
    {
        c.sin2k[i] = 2.0f * c.cos2k[i - 1] * c.sin2k[i - 1];
        c.cos2k[i] = sqrf_(c.cos2k[i - 1]) - sqrf_(c.sin2k[i - 1]);
    }
    return c;
}
struct HairTerms
{
    float h, eta, diffuse_w;
    v3 sigma_a, tint;
    const HairConst* c; // (read where it is used: 24 per-material values the kernel does not have to hold in registers across the BSDF)
};
SKH_DI HairTerms hair_terms(const Material& m, const HairConst& hc)
{
    HairTerms t;
    t.h = 2.0f * SKH_HAIR_TEXCOORD_Y - 1.0f;
    t.eta = m.ior > 1.0f ? m.ior : 1.55f;
    t.sigma_a = mk3(fmaxf(m.reserved[0], 0.0f), fmaxf(m.reserved[1], 0.0f), fmaxf(m.reserved[2], 0.0f));
    t.tint = mk3(m.base_color[0], m.base_color[1], m.base_color[2]);
    t.diffuse_w = clampf(m.reserved[5], 0.0f, 1.0f);
    t.c = &hc;
    return t;
}
SKH_DI void hair_Ap(const HairTerms& t, float cosThetaO, const v3& T, v3 ap[4], float apPdf[4])
{
    const float cosGammaO = safe_sqrtf_(1.0f - t.h * t.h);
    const float cosTheta = cosThetaO * cosGammaO;
    float cost;
    const float f = fresnel_dielectric(clampf(cosTheta, 0.0f, 1.0f), 1.0f / t.eta, cost);
    ap[0] = mk3(f);
    ap[1] = T * sqrf_(1.0f - f);
    ap[2] = ap[1] * T * f;
    ap[3] = (ap[2] * f) * T / (mk3(1.0f) - T * f);
    float y[4], sum = 0.0f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
    {
        y[p] = lum3(ap[p]);
        sum += y[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
        apPdf[p] = sum > 0.0f ? y[p] / sum : 0.25f;
}
struct HairGeom
{
    float sinThetaO, cosThetaO, phiO, gammaO, gammaT;
    v3 T;
};
SKH_DI HairGeom hair_geom(const HairTerms& t, const v3& wo)
{
    HairGeom g;
    g.sinThetaO = clampf(wo.x, -1.0f, 1.0f);
    g.cosThetaO = safe_sqrtf_(1.0f - sqrf_(g.sinThetaO));
    g.phiO = skm::atan2f_(wo.z, wo.y);
    const float sinThetaT = g.sinThetaO / t.eta;
    const float cosThetaT = safe_sqrtf_(1.0f - sqrf_(sinThetaT));
    const float etap = sqrtf(t.eta * t.eta - sqrf_(g.sinThetaO)) / fmaxf(g.cosThetaO, 1e-6f);
    const float sinGammaT = t.h / etap;
    const float cosGammaT = safe_sqrtf_(1.0f - sqrf_(sinGammaT));
    g.gammaT = safe_asinf_(sinGammaT);
    g.gammaO = safe_asinf_(t.h);
    const float l = 2.0f * cosGammaT / fmaxf(cosThetaT, 1e-6f);
    g.T = mk3(skm::expf_(-t.sigma_a.x * l), skm::expf_(-t.sigma_a.y * l), skm::expf_(-t.sigma_a.z * l));
    return g;
}
SKH_DI void hair_tilt(const HairTerms& t, const HairGeom& g, int p, float& sinThetaOp, float& cosThetaOp)
{
    if (p == 0)
    {
        sinThetaOp = g.sinThetaO * t.c->cos2k[1] - g.cosThetaO * t.c->sin2k[1];
        cosThetaOp = g.cosThetaO * t.c->cos2k[1] + g.sinThetaO * t.c->sin2k[1];
    }
    else if (p == 1)
    {
        sinThetaOp = g.sinThetaO * t.c->cos2k[0] + g.cosThetaO * t.c->sin2k[0];
        cosThetaOp = g.cosThetaO * t.c->cos2k[0] - g.sinThetaO * t.c->sin2k[0];
    }
    else if (p == 2)
    {
        sinThetaOp = g.sinThetaO * t.c->cos2k[2] + g.cosThetaO * t.c->sin2k[2];
        cosThetaOp = g.cosThetaO * t.c->cos2k[2] - g.sinThetaO * t.c->sin2k[2];
    }
    else
    {
        sinThetaOp = g.sinThetaO;
        cosThetaOp = g.cosThetaO;
    }
    cosThetaOp = fabsf(cosThetaOp);
}
// (the _g forms take the outgoing direction's geometry and the attenuations ready-made: one hit of a hair material evaluates the BSDF twice for the same k1 --
// the sampled direction and the light sample's -- and k_shade's hair path shares them: hair_sample_and_evaluate)
SKH_DI void hair_eval_local_g(const HairTerms& t, const HairGeom& g, const v3 ap[4], const float apPdf[4], const v3& wi, v3& f_cos, float& pdf)
{
    const float sinThetaI = clampf(wi.x, -1.0f, 1.0f);
    const float cosThetaI = safe_sqrtf_(1.0f - sqrf_(sinThetaI));
    const float phi = skm::atan2f_(wi.z, wi.y) - g.phiO;
    f_cos = mk3(0.0f);
    pdf = 0.0f;
#pragma unroll
    for (int p = 0; p < 3; ++p)
    {
        float so, co;
        hair_tilt(t, g, p, so, co);
        const float mn = hair_Mp(cosThetaI, co, sinThetaI, so, t.c->v[p], t.c->invV[p], t.c->log1over2v[p], t.c->sinhDen[p]) * hair_Np(phi, p, *t.c, g.gammaO, g.gammaT);
        f_cos = f_cos + ap[p] * mn;
        pdf += apPdf[p] * mn;
    }
    const float mr = hair_Mp(cosThetaI, g.cosThetaO, sinThetaI, g.sinThetaO, t.c->v[2], t.c->invV[2], t.c->log1over2v[2], t.c->sinhDen[2]) * (1.0f / (2.0f * SKH_PI));
    f_cos = f_cos + ap[3] * mr;
    pdf += apPdf[3] * mr;
}
SKH_DI void hair_eval_local(const HairTerms& t, const v3& wo, const v3& wi, v3& f_cos, float& pdf)
{
    const HairGeom g = hair_geom(t, wo);
    v3 ap[4];
    float apPdf[4];
    hair_Ap(t, g.cosThetaO, g.T, ap, apPdf);
    hair_eval_local_g(t, g, ap, apPdf, wi, f_cos, pdf);
}
SKH_DI v3 hair_sample_local_g(const HairTerms& t, const HairGeom& g, const float apPdf[4], float u0, float u1, float u2, float u3)
{
    int p = 0;
    float u = u2;
    for (; p < 3; ++p)
    {
        if (u < apPdf[p])
            break;
        u -= apPdf[p];
    }
    float so, co;
    hair_tilt(t, g, p, so, co);
    const float vp = p == 0 ? t.c->v[0] : (p == 1 ? t.c->v[1] : t.c->v[2]); // (the residual lobe uses TRT's)
    const float em = p == 0 ? t.c->expm2overV[0] : (p == 1 ? t.c->expm2overV[1] : t.c->expm2overV[2]);
    const float ua = fmaxf(u0, 1e-5f);
    const float cosTheta = 1.0f + vp * skm::logf_(ua + (1.0f - ua) * em);
    const float sinTheta = safe_sqrtf_(1.0f - sqrf_(cosTheta));
    const float cosPhi = skm::cosf_(2.0f * SKH_PI * u1);
    const float sinThetaI = -cosTheta * so + sinTheta * cosPhi * co;
    const float cosThetaI = safe_sqrtf_(1.0f - sqrf_(sinThetaI));
    const float dphi = p < 3 ? hair_Phi(p, g.gammaO, g.gammaT) + hair_sample_trimmed_logistic(u3, *t.c, -SKH_PI, SKH_PI) : 2.0f * SKH_PI * u3;
    const float phiI = g.phiO + dphi;
    return mk3(sinThetaI, cosThetaI * skm::cosf_(phiI), cosThetaI * skm::sinf_(phiI));
}
SKH_DI v3 hair_sample_local(const HairTerms& t, const v3& wo, float u0, float u1, float u2, float u3)
{
    const HairGeom g = hair_geom(t, wo);
    v3 ap[4];
    float apPdf[4];
    hair_Ap(t, g.cosThetaO, g.T, ap, apPdf);
    return hair_sample_local_g(t, g, apPdf, u0, u1, u2, u3);
}
SKH_DI bool hair_frame(const v3& normal, const v3& tangent_u, v3& X, v3& Y, v3& Z)
{
    const float tl = dot(tangent_u, tangent_u);
    if (!(tl > 0.0f))
        return false;
    X = tangent_u * (1.0f / sqrtf(tl));
    const v3 z = normal - X * dot(normal, X);
    const float zl = dot(z, z);
    if (!(zl > 1e-12f))
        return false;
    Z = z * (1.0f / sqrtf(zl));
    Y = cross(Z, X);
    return true;
}

// mdlcode_sample equivalent; `inside` selects ior1/ior2 as closest_hit.cu:496-498.  HAIR: the build that carries the hair
// distribution function (scenes without a hair material run the build without it: fewer registers).
template <bool HAIR>
SKH_DI void bsdf_sample(const Material& m, const v3& stN, const v3& stNg, const v3& stT, const v3& k1, float xi0, float xi1, float xi2,
                        float xi3, bool inside, BsdfSample& out, const HairConst* hc = nullptr /* (HAIR) the material's constants: DevScene::hairConst */)
{
    v3 N = stN, Ng = stNg;
    if (dot(Ng, k1) < 0.0f)
    {
        N = -N;
        Ng = -Ng;
    }
    v3 b1, b2;
    onb_from_z(N, b1, b2);
    const v3 wo = mk3(dot(k1, b1), dot(k1, b2), dot(k1, N));
    out.k2 = mk3(0.0f);
    out.bsdf_over_pdf = mk3(0.0f);
    out.pdf = 0.0f;
    out.event_type = EV_ABSORB;
    const v3 base = mk3(m.base_color[0], m.base_color[1], m.base_color[2]);
    if (m.type == 0)
    {
        float cosT;
        const v3 w = cosine_hemisphere(xi0, xi1, cosT);
        const v3 k2 = normalize(w.x * b1 + w.y * b2 + w.z * N);
        if (cosT <= 0.0f || dot(k2, Ng) <= 0.0f)
            return;
        out.k2 = k2;
        out.pdf = cosT / SKH_PI;
        out.bsdf_over_pdf = base;
        out.event_type = EV_DIFFUSE | EV_REFLECTION;
        return;
    }
    if (HAIR && m.type == 3)
    {
        v3 X, Y, Z;
        if (!hair_frame(stN, stT, X, Y, Z))
            return;
        const HairTerms t = hair_terms(m, *hc);
        const v3 ho = mk3(dot(k1, X), dot(k1, Y), dot(k1, Z));
        float u2 = xi2;
        if (u2 < t.diffuse_w)
        {
            float cosT;
            const v3 w = cosine_hemisphere(xi0, xi1, cosT);
            const v3 k2 = normalize(w.x * b1 + w.y * b2 + w.z * N);
            if (cosT <= 0.0f)
                return;
            const v3 hi = mk3(dot(k2, X), dot(k2, Y), dot(k2, Z));
            v3 fh;
            float ph;
            hair_eval_local(t, ho, hi, fh, ph);
            const float pd = cosT / SKH_PI;
            const float pdf = t.diffuse_w * pd + (1.0f - t.diffuse_w) * ph;
            out.k2 = k2;
            out.pdf = pdf;
            out.bsdf_over_pdf = (t.tint * (t.diffuse_w * pd) + fh * (1.0f - t.diffuse_w)) / pdf;
            out.event_type = EV_DIFFUSE | EV_REFLECTION;
            return;
        }
        u2 = (u2 - t.diffuse_w) / (1.0f - t.diffuse_w);
        const v3 hi = hair_sample_local(t, ho, xi0, xi1, u2, xi3);
        v3 fh;
        float ph;
        hair_eval_local(t, ho, hi, fh, ph);
        const v3 k2 = normalize(hi.x * X + hi.y * Y + hi.z * Z);
        const float cosN = dot(k2, N);
        const float pd = cosN > 0.0f ? cosN / SKH_PI : 0.0f;
        const float pdf = t.diffuse_w * pd + (1.0f - t.diffuse_w) * ph;
        if (!(pdf > 0.0f) || !(ph > 0.0f))
            return;
        out.k2 = k2;
        out.pdf = pdf;
        out.bsdf_over_pdf = (t.tint * (t.diffuse_w * pd) + fh * (1.0f - t.diffuse_w)) / pdf;
        out.event_type = EV_GLOSSY | EV_REFLECTION;
        return;
    }
    if (m.type == 1)
    {
        if (wo.z <= 0.0f)
            return;
        const PbrTerms t = pbr_terms(m);
        v3 wi;
        int ev;
        if (xi2 < t.p_spec)
        {
            const v3 h = ggx_sample_vndf(wo, t.alpha, xi0, xi1);
            wi = h * (2.0f * dot(wo, h)) - wo;
            ev = EV_GLOSSY | EV_REFLECTION;
        }
        else
        {
            float cosT;
            wi = cosine_hemisphere(xi0, xi1, cosT);
            ev = EV_DIFFUSE | EV_REFLECTION;
        }
        if (wi.z <= 0.0f)
            return;
        const v3 k2 = normalize(wi.x * b1 + wi.y * b2 + wi.z * N);
        if (dot(k2, Ng) <= 0.0f)
            return;
        v3 fd, fs;
        float pdf;
        pbr_eval_local(t, wo, wi, fd, fs, pdf);
        if (!(pdf > 0.0f))
            return;
        out.k2 = k2;
        out.pdf = pdf;
        out.bsdf_over_pdf = (fd + fs) / pdf;
        out.event_type = ev;
        return;
    }
    if (m.type == 2)
    {
        const float n1 = inside ? m.ior : 1.0f;
        const float n2 = inside ? 1.0f : m.ior;
        const float eta = n1 / n2;
        if (m.roughness >= SKH_GLASS_SMOOTH_BELOW)
        {
            if (wo.z <= 0.0f)
                return;
            const float alpha = fmaxf(m.roughness * m.roughness, 1e-4f);
            v3 wi, weight;
            float pdf;
            bool transmitted;
            if (!rough_glass_sample_local(alpha, eta, base, wo, xi0, xi1, xi2, wi, weight, pdf, transmitted))
                return;
            out.k2 = normalize(wi.x * b1 + wi.y * b2 + wi.z * N);
            out.pdf = pdf;
            out.bsdf_over_pdf = weight;
            out.event_type = EV_GLOSSY | (transmitted ? EV_TRANSMISSION : EV_REFLECTION);
            return;
        }
        const float cosi = fminf(fmaxf(wo.z, 0.0f), 1.0f);
        float cost;
        const float F = fresnel_dielectric(cosi, eta, cost);
        if (xi2 < F)
        {
            out.k2 = normalize(N * (2.0f * dot(k1, N)) - k1);
            out.bsdf_over_pdf = mk3(1.0f);
            out.event_type = EV_SPECULAR | EV_REFLECTION;
        }
        else
        {
            out.k2 = normalize(N * (eta * cosi - cost) - k1 * eta);
            out.bsdf_over_pdf = base;
            out.event_type = EV_SPECULAR | EV_TRANSMISSION;
        }
        out.pdf = 0.0f;
        return;
    }
}
// One hit of a HAIR material in k_shade's hair build: mdlcode_sample (what bsdf_sample<true> returns for m.type == 3) and, when the light sample's direction k2e is
// worth evaluating, mdlcode_evaluate for it (what bsdf_evaluate<true> returns) -- the two calls of rounds 2-5 with the fibre frame, the outgoing direction's
// geometry (atan2, two asin, three exp) and the attenuations A_p computed ONCE: the same operations on the same operands, so the same bits.
SKH_DI void hair_sample_and_evaluate(const Material& m, const v3& stN, const v3& stNg, const v3& stT, const v3& k1, float xi0, float xi1, float xi2, float xi3,
                                     const HairConst* hc, BsdfSample& out, bool wantEval, const v3& k2e, BsdfEval& ev)
{
    v3 N = stN, Ng = stNg;
    if (dot(Ng, k1) < 0.0f)
    {
        N = -N;
        Ng = -Ng;
    }
    out.k2 = mk3(0.0f);
    out.bsdf_over_pdf = mk3(0.0f);
    out.pdf = 0.0f;
    out.event_type = EV_ABSORB;
    ev.bsdf_diffuse = mk3(0.0f);
    ev.bsdf_glossy = mk3(0.0f);
    ev.pdf = 0.0f;
    v3 X, Y, Z;
    if (!hair_frame(stN, stT, X, Y, Z))
        return;
    const HairTerms t = hair_terms(m, *hc);
    const v3 ho = mk3(dot(k1, X), dot(k1, Y), dot(k1, Z));
    const HairGeom g = hair_geom(t, ho);
    v3 ap[4];
    float apPdf[4];
    hair_Ap(t, g.cosThetaO, g.T, ap, apPdf);
    if (wantEval)
    {
        const v3 hi = mk3(dot(k2e, X), dot(k2e, Y), dot(k2e, Z));
        v3 fh;
        float ph;
        hair_eval_local_g(t, g, ap, apPdf, hi, fh, ph);
        const float cosN = dot(k2e, N);
        const float pd = cosN > 0.0f ? cosN / SKH_PI : 0.0f;
        ev.bsdf_glossy = fh * (1.0f - t.diffuse_w);
        ev.bsdf_diffuse = t.tint * (t.diffuse_w * pd);
        ev.pdf = t.diffuse_w * pd + (1.0f - t.diffuse_w) * ph;
    }
    float u2 = xi2;
    if (u2 < t.diffuse_w)
    {
        v3 b1, b2;
        onb_from_z(N, b1, b2);
        float cosT;
        const v3 w = cosine_hemisphere(xi0, xi1, cosT);
        const v3 k2 = normalize(w.x * b1 + w.y * b2 + w.z * N);
        if (cosT <= 0.0f)
            return;
        const v3 hi = mk3(dot(k2, X), dot(k2, Y), dot(k2, Z));
        v3 fh;
        float ph;
        hair_eval_local_g(t, g, ap, apPdf, hi, fh, ph);
        const float pd = cosT / SKH_PI;
        const float pdf = t.diffuse_w * pd + (1.0f - t.diffuse_w) * ph;
        out.k2 = k2;
        out.pdf = pdf;
        out.bsdf_over_pdf = (t.tint * (t.diffuse_w * pd) + fh * (1.0f - t.diffuse_w)) / pdf;
        out.event_type = EV_DIFFUSE | EV_REFLECTION;
        return;
    }
    u2 = (u2 - t.diffuse_w) / (1.0f - t.diffuse_w);
    const v3 hi = hair_sample_local_g(t, g, apPdf, xi0, xi1, u2, xi3);
    v3 fh;
    float ph;
    hair_eval_local_g(t, g, ap, apPdf, hi, fh, ph);
    const v3 k2 = normalize(hi.x * X + hi.y * Y + hi.z * Z);
    const float cosN = dot(k2, N);
    const float pd = cosN > 0.0f ? cosN / SKH_PI : 0.0f;
    const float pdf = t.diffuse_w * pd + (1.0f - t.diffuse_w) * ph;
    if (!(pdf > 0.0f) || !(ph > 0.0f))
        return;
    out.k2 = k2;
    out.pdf = pdf;
    out.bsdf_over_pdf = (t.tint * (t.diffuse_w * pd) + fh * (1.0f - t.diffuse_w)) / pdf;
    out.event_type = EV_GLOSSY | EV_REFLECTION;
}

// mdlcode_evaluate equivalent
template <bool HAIR>
SKH_DI void bsdf_evaluate(const Material& m, const v3& stN, const v3& stNg, const v3& stT, const v3& k1, const v3& k2, bool inside, BsdfEval& out,
                          const HairConst* hc = nullptr)
{
    v3 N = stN, Ng = stNg;
    if (dot(Ng, k1) < 0.0f)
    {
        N = -N;
        Ng = -Ng;
    }
    out.bsdf_diffuse = mk3(0.0f);
    out.bsdf_glossy = mk3(0.0f);
    out.pdf = 0.0f;
    const v3 base = mk3(m.base_color[0], m.base_color[1], m.base_color[2]);
    if (m.type == 0)
    {
        const float nk2 = dot(N, k2);
        if (nk2 <= 0.0f || dot(Ng, k2) <= 0.0f)
            return;
        out.bsdf_diffuse = base * (nk2 / SKH_PI);
        out.pdf = nk2 / SKH_PI;
        return;
    }
    if (HAIR && m.type == 3)
    {
        v3 X, Y, Z;
        if (!hair_frame(stN, stT, X, Y, Z))
            return;
        const HairTerms t = hair_terms(m, *hc);
        const v3 ho = mk3(dot(k1, X), dot(k1, Y), dot(k1, Z));
        const v3 hi = mk3(dot(k2, X), dot(k2, Y), dot(k2, Z));
        v3 fh;
        float ph;
        hair_eval_local(t, ho, hi, fh, ph);
        const float cosN = dot(k2, N);
        const float pd = cosN > 0.0f ? cosN / SKH_PI : 0.0f;
        out.bsdf_glossy = fh * (1.0f - t.diffuse_w);
        out.bsdf_diffuse = t.tint * (t.diffuse_w * pd);
        out.pdf = t.diffuse_w * pd + (1.0f - t.diffuse_w) * ph;
        return;
    }
    if (m.type == 1)
    {
        v3 b1, b2;
        onb_from_z(N, b1, b2);
        const v3 wo = mk3(dot(k1, b1), dot(k1, b2), dot(k1, N));
        const v3 wi = mk3(dot(k2, b1), dot(k2, b2), dot(k2, N));
        if (dot(Ng, k2) <= 0.0f)
            return;
        const PbrTerms t = pbr_terms(m);
        pbr_eval_local(t, wo, wi, out.bsdf_diffuse, out.bsdf_glossy, out.pdf);
        return;
    }
    if (m.type == 2 && m.roughness >= SKH_GLASS_SMOOTH_BELOW)
    {
        v3 b1, b2;
        onb_from_z(N, b1, b2);
        const v3 wo = mk3(dot(k1, b1), dot(k1, b2), dot(k1, N));
        const v3 wi = mk3(dot(k2, b1), dot(k2, b2), dot(k2, N));
        const float n1 = inside ? m.ior : 1.0f;
        const float n2 = inside ? 1.0f : m.ior;
        rough_glass_eval_local(fmaxf(m.roughness * m.roughness, 1e-4f), n1 / n2, base, wo, wi, out.bsdf_glossy, out.pdf);
    }
}

} // namespace skh
