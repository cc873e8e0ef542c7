// ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the arithmetic on Strelka's render() hot path.
// Nothing in the product (strelka_amd/, the C-ABI library) may include, link or call this file.
//
// ork_math.h: float3/float4 helpers with the exact operator semantics of the reference's sutil/vec_math.h
// (NVIDIA SDK helper).  The semantics that matter for bit-level agreement are restated, not copied:
//   * vector / scalar  == vector * (1.0f / scalar)          (sutil/vec_math.h:487-491)
//   * normalize(v)     == v * (1.0f / sqrtf(dot(v,v)))      (sutil/vec_math.h:549-553)
//   * dot(a,b)         == a.x*b.x + a.y*b.y + a.z*b.z  (left to right)   (sutil/vec_math.h:530-533)
//   * lerp(a,b,t)      == a + t*(b-a)                       (sutil/vec_math.h:504-507)
// Compile with -ffp-contract=off so no FMA contraction changes the rounding.
#pragma once
// sin / cos / acos / asin / atan2 / exp / log / sinh / pow: the SAME text the device code compiles (fixed polynomials in correctly rounded
// operations: identical bits on both sides; accuracy against glibc: tests/test_libm.py).  The one include that points from the checker to the
// product tree -- for libm stand-ins only; everything that restates the reference is written separately in oracle/.
#include "../strelka_amd/csrc/skh_libm.h"
#include <cmath>
#include <cstdint>
#include <cstring>

// -DORK_LIBM_GLIBC (oracle/Makefile target liboracle_glibc.so): a SECOND build of the checker whose transcendentals are glibc's, not the
// shared text -- the second opinion on skh_libm.h inside whole renders (tests/test_oracle_render.py, tests/test_gpu_parity.py hold the
// default build and the GPU against it at round 4's image tolerances: an error in a shared polynomial would show there as more than
// last-ulp path flips).  Every `skm::` below this line then names these wrappers.
#ifdef ORK_LIBM_GLIBC
namespace skm_glibc
{
static inline float sinf_(float x) { return ::sinf(x); }
static inline float cosf_(float x) { return ::cosf(x); }
static inline float asinf_(float x) { return ::asinf(x); }
static inline float acosf_(float x) { return ::acosf(x); }
static inline float atan2f_(float y, float x) { return ::atan2f(y, x); }
static inline float expf_(float x) { return ::expf(x); }
static inline float logf_(float x) { return ::logf(x); }
static inline float sinhf_(float x) { return ::sinhf(x); }
static inline float powf_(float x, float y) { return ::powf(x, y); }
} // namespace skm_glibc
#define skm skm_glibc
#endif

namespace ork
{

struct f2
{
    float x, y;
};
struct f3
{
    float x, y, z;
};
struct f4
{
    float x, y, z, w;
};

static inline f3 mk3(float x, float y, float z)
{
    return f3{ x, y, z };
}
static inline f3 mk3(float s)
{
    return f3{ s, s, s };
}
static inline f3 mk3(const f4& a)
{
    return f3{ a.x, a.y, a.z };
}
static inline f4 mk4(const f3& a, float w)
{
    return f4{ a.x, a.y, a.z, w };
}

static inline f3 operator+(const f3& a, const f3& b)
{
    return f3{ a.x + b.x, a.y + b.y, a.z + b.z };
}
static inline f3 operator-(const f3& a, const f3& b)
{
    return f3{ a.x - b.x, a.y - b.y, a.z - b.z };
}
static inline f3 operator-(const f3& a)
{
    return f3{ -a.x, -a.y, -a.z };
}
static inline f3 operator*(const f3& a, const f3& b)
{
    return f3{ a.x * b.x, a.y * b.y, a.z * b.z };
}
static inline f3 operator*(const f3& a, float s)
{
    return f3{ a.x * s, a.y * s, a.z * s };
}
static inline f3 operator*(float s, const f3& a)
{
    return f3{ s * a.x, s * a.y, s * a.z };
}
static inline f3 operator/(const f3& a, const f3& b)
{
    return f3{ a.x / b.x, a.y / b.y, a.z / b.z };
}
static inline f3 operator/(const f3& a, float s)
{
    const float inv = 1.0f / s;
    return a * inv;
}
static inline f3& operator+=(f3& a, const f3& b)
{
    a = a + b;
    return a;
}
static inline f3& operator-=(f3& a, const f3& b)
{
    a = a - b;
    return a;
}
static inline f3& operator*=(f3& a, const f3& b)
{
    a = a * b;
    return a;
}
static inline f3& operator*=(f3& a, float s)
{
    a = a * s;
    return a;
}

static inline f4 operator+(const f4& a, const f4& b)
{
    return f4{ a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w };
}
static inline f4 operator-(const f4& a, const f4& b)
{
    return f4{ a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w };
}
static inline f4 operator*(const f4& a, float s)
{
    return f4{ a.x * s, a.y * s, a.z * s, a.w * s };
}
static inline f4 operator*(float s, const f4& a)
{
    return f4{ s * a.x, s * a.y, s * a.z, s * a.w };
}
static inline f4 operator/(const f4& a, float s)
{
    const float inv = 1.0f / s;
    return a * inv;
}

static inline float dot(const f3& a, const f3& b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
static inline f3 cross(const f3& a, const f3& b)
{
    return f3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
static inline float length(const f3& v)
{
    return sqrtf(dot(v, v));
}
static inline f3 normalize(const f3& v)
{
    const float invLen = 1.0f / sqrtf(dot(v, v));
    return v * invLen;
}
static inline float clampf(float f, float a, float b)
{
    return fmaxf(a, fminf(f, b)); // sutil/vec_math.h clamp(float): max(a, min(f, b))
}
static inline float saturatef(float v)
{
    return clampf(v, 0.0f, 1.0f);
}
static inline f3 lerp3(const f3& a, const f3& b, float t)
{
    return a + t * (b - a);
}
static inline bool all3(const f3& v) // sutil/vec_math_adv.h:39-42
{
    return v.x != 0.0f && v.y != 0.0f && v.z != 0.0f;
}
static inline bool isnan3(const f3& v)
{
    return std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z);
}

static inline uint32_t f2u(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float u2f(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline int32_t f2i(float f)
{
    int32_t i;
    memcpy(&i, &f, 4);
    return i;
}
static inline float i2f(int32_t i)
{
    float f;
    memcpy(&f, &i, 4);
    return f;
}

// row-major 4x4 times float4 -- sutil/Matrix.h:467-488 (each row: m0*x + m1*y + m2*z + m3*w, left to right)
static inline f4 mul44(const float* m, const f4& v)
{
    f4 r;
    r.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w;
    r.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * v.w;
    r.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * v.w;
    r.w = m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15] * v.w;
    return r;
}

// Affine 3x4 row-major transforms.  Point: ((m0*x + m1*y) + m2*z) + m3.  Vector: (m0*x + m1*y) + m2*z.
// This is the order optixTransformPointFromObjectToWorldSpace-style helpers are specified to here; the
// HIP kernels use the same order so that object-space rays agree bit for bit.
static inline f3 xform_point(const float* m, const f3& p)
{
    return f3{ ((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
               ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11] };
}
// world -> object for points, better conditioned than w2o * p: R^-1 (p - T) with R^-1 = the 3x3 of w2o and T = the translation of
// the object-to-world transform (same definition, same order as the product's xform_point_rel)
static inline f3 xform_point_rel(const float* w2o, const float* o2w, const f3& p)
{
    const float x = p.x - o2w[3], y = p.y - o2w[7], z = p.z - o2w[11];
    return f3{ (w2o[0] * x + w2o[1] * y) + w2o[2] * z, (w2o[4] * x + w2o[5] * y) + w2o[6] * z, (w2o[8] * x + w2o[9] * y) + w2o[10] * z };
}
static inline f3 xform_vector(const float* m, const f3& v)
{
    return f3{ (m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
               (m[8] * v.x + m[9] * v.y) + m[10] * v.z };
}
// normal transform object->world = transpose(world_to_object 3x3) * n
static inline f3 xform_normal(const float* w2o, const f3& n)
{
    return f3{ (w2o[0] * n.x + w2o[4] * n.y) + w2o[8] * n.z, (w2o[1] * n.x + w2o[5] * n.y) + w2o[9] * n.z,
               (w2o[2] * n.x + w2o[6] * n.y) + w2o[10] * n.z };
}

// Inverse of an affine 3x4 (row-major), computed in fp64 by the adjugate, rounded once to fp32.
// (The reference takes glm::inverse(instance.transform): OptixRender.cpp:793-795; OptiX derives its own
// world->object internally.)  The product computes it with the same formula in the same order.
static inline void invert_affine(const float* m, float* out)
{
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double tx = m[3], ty = m[7], tz = m[11];
    const double A = e * i - f * h;
    const double B = c * h - b * i;
    const double C = b * f - c * e;
    const double D = f * g - d * i;
    const double E = a * i - c * g;
    const double F = c * d - a * f;
    const double G = d * h - e * g;
    const double H = b * g - a * h;
    const double I = a * e - b * d;
    const double det = a * A + b * D + c * G;
    const double r = 1.0 / det;
    const double i00 = A * r, i01 = B * r, i02 = C * r;
    const double i10 = D * r, i11 = E * r, i12 = F * r;
    const double i20 = G * r, i21 = H * r, i22 = I * r;
    out[0] = (float)i00;
    out[1] = (float)i01;
    out[2] = (float)i02;
    out[3] = (float)(-(i00 * tx + i01 * ty + i02 * tz));
    out[4] = (float)i10;
    out[5] = (float)i11;
    out[6] = (float)i12;
    out[7] = (float)(-(i10 * tx + i11 * ty + i12 * tz));
    out[8] = (float)i20;
    out[9] = (float)i21;
    out[10] = (float)i22;
    out[11] = (float)(-(i20 * tx + i21 * ty + i22 * tz));
}

} // namespace ork
