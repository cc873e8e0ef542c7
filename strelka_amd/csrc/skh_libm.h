// skh_libm.h -- the transcendental functions of the render path, written once in IEEE-754 single-precision + - * / sqrt and fma.
//
// Why: the reference calls sinf / cosf / acosf / expf / logf / ... of whatever libm it is built against (CUDA's on the OptiX side).  Between
// glibc (the CPU oracle) and the ROCm device library those functions differ in the last ulps, and a 1-ulp direction change moves a path
// across a triangle edge: the only thing that kept HIP-vs-oracle IMAGES at a tolerance while hit records, samplers and accumulation were
// already bit-exact.  Every operation below is correctly rounded on both sides (v_fma_f32 / v_sqrt_f32 with the
// compiler's correction sequence / IEEE division on gfx950; SSE / FMA3 on the host), both sides compile with -ffp-contract=off, and the
// polynomials are fixed: the functions return THE SAME BITS on the CPU and on the GPU.  Accuracy, measured against glibc in double
// precision over 2 x 10^7 arguments per function and range (tests/test_libm.py keeps the bars): sin / cos <= 1.6 ulp for |x| <= 400,
// acos <= 1.3, asin <= 2.4, exp / log <= 1.1, sinh <= 1.7, atan2 <= 3.2 ulp, pow(x, 2.2) and pow(x, 1 / 2.2) <= 3.8 ulp on [1e-6, 1e4] -- inside
// the <= 4 ulp bars at which the reference-generated fixtures (tests/golden/lights_*.f32) were already held, which is what "same
// algorithm as the reference" can mean for a libm call.
//
// Included by skh_device.h (device code) and -- the one dependency that points from the CPU checker to the product tree, and only for
// these libm stand-ins -- by the checker's math header.  Polynomial forms and coefficients: the classic Cephes single-precision ones
// (Moshier), evaluated with fma.
#pragma once
#include <stdint.h>

#ifndef SKH_LIBM_FN
#    if defined(__HIPCC__)
#        define SKH_LIBM_FN static __host__ __device__ __forceinline__
#    else
#        define SKH_LIBM_FN static inline
#    endif
#endif

namespace skm
{

SKH_LIBM_FN uint32_t f2u(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
SKH_LIBM_FN float u2f(uint32_t u)
{
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
SKH_LIBM_FN float fabs_(float x)
{
    return u2f(f2u(x) & 0x7fffffffu);
}
SKH_LIBM_FN float copysign_(float mag, float sgn)
{
    return u2f((f2u(mag) & 0x7fffffffu) | (f2u(sgn) & 0x80000000u));
}
SKH_LIBM_FN bool isnan_(float x)
{
    return (f2u(x) & 0x7fffffffu) > 0x7f800000u;
}
// x * 2^k, exact while the result is normal; two-step for results in the subnormal range
SKH_LIBM_FN float scale2(float x, int k)
{
    if (k >= -126 && k <= 127)
        return x * u2f((uint32_t)(k + 127) << 23);
    if (k > 127)
    {
        x = x * u2f(0x7f000000u); // 2^127
        k -= 127;
        if (k > 127)
            k = 127;
    }
    else if (k < -126)
    {
        x = x * u2f(0x0c800000u); // 2^-102
        k += 102;
        if (k < -126)
            k = -126;
    }
    return x * u2f((uint32_t)(k + 127) << 23);
}

// ---- sin / cos --------------------------------------------------------------------------------------------------------------------
// k = nearest integer to x * 2/pi; r = x - k * pi/2 with pi/2 in three parts (Cody-Waite with fma: exact products), |r| <= pi/4 (+ a hair);
// then the two minimax polynomials, selected and signed by k mod 4.  Arguments beyond ~1e5 lose accuracy gradually (no Payne-Hanek), never
// determinism.
SKH_LIBM_FN void sincos_reduce(float x, float& r, int& q)
{
    const float kf = __builtin_rintf(x * 0.636619772367581343f); // (round-to-nearest-even conversion: v_rndne_f32 / roundss, exact)
    q = (int)(kf - 4.0f * __builtin_floorf(kf * 0.25f)); // k mod 4 in exact float steps: a huge k must not reach a float -> int conversion (its overflow differs between the two sides)
    r = __builtin_fmaf(-kf, 1.57079625129699707031e+0f, x);
    r = __builtin_fmaf(-kf, 7.54978941586159635335e-8f, r);
    r = __builtin_fmaf(-kf, 5.39030285815811843e-15f, r);
}
SKH_LIBM_FN float sin_poly(float r)
{
    const float z = r * r;
    float p = -1.9515295891e-4f;
    p = __builtin_fmaf(p, z, 8.3321608736e-3f);
    p = __builtin_fmaf(p, z, -1.6666654611e-1f);
    return __builtin_fmaf(p * z, r, r);
}
SKH_LIBM_FN float cos_poly(float r)
{
    const float z = r * r;
    float p = 2.443315711809948e-5f;
    p = __builtin_fmaf(p, z, -1.388731625493765e-3f);
    p = __builtin_fmaf(p, z, 4.166664568298827e-2f);
    return __builtin_fmaf(p * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
}
SKH_LIBM_FN float sinf_(float x)
{
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
    return __ocml_sin_f32(x);
#endif
    if (!(fabs_(x) < 3.0e38f))
        return x - x; // inf, nan -> nan
    float r;
    int q;
    sincos_reduce(x, r, q);
    const float v = (q & 1) ? cos_poly(r) : sin_poly(r);
    return (q & 2) ? -v : v;
}
SKH_LIBM_FN float cosf_(float x)
{
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
    return __ocml_cos_f32(x);
#endif
    if (!(fabs_(x) < 3.0e38f))
        return x - x;
    float r;
    int q;
    sincos_reduce(x, r, q);
    const float v = (q & 1) ? sin_poly(r) : cos_poly(r);
    return ((q + 1) & 2) ? -v : v;
}

// ---- asin / acos ------------------------------------------------------------------------------------------------------------------
// asin(t) = t + t^3 P(t^2) on [0, 0.5]; beyond, asin(x) = pi/2 - 2 asin(sqrt((1 - x) / 2))
SKH_LIBM_FN float asin_poly(float z)
{
    float p = 4.2163199048e-2f;
    p = __builtin_fmaf(p, z, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z, 1.6666752422e-1f);
    return p;
}
SKH_LIBM_FN float asinf_(float x)
{
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
    return __ocml_asin_f32(x);
#endif
    const float a = fabs_(x);
    if (!(a <= 1.0f))
        return (x - x) / (x - x); // nan
    float r;
    if (a > 0.5f)
    {
        const float z = 0.5f * (1.0f - a);
        const float s = __builtin_sqrtf(z);
        const float t = __builtin_fmaf(s * z, asin_poly(z), s);
        r = 1.57079637050628662f - (2.0f * t - -4.37113882867379289e-8f); // pi/2 = hi + lo: the small term joins t, one rounding at the result's size
    }
    else
    {
        const float z = a * a;
        r = __builtin_fmaf(a * z, asin_poly(z), a);
    }
    return copysign_(r, x);
}
SKH_LIBM_FN float acosf_(float x)
{
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
    return __ocml_acos_f32(x);
#endif
    const float a = fabs_(x);
    if (!(a <= 1.0f))
        return (x - x) / (x - x);
    if (a > 0.5f)
    {
        const float z = 0.5f * (1.0f - a);
        const float s = __builtin_sqrtf(z);
        const float t = 2.0f * __builtin_fmaf(s * z, asin_poly(z), s);
        // pi as hi + lo so that pi - t keeps its last bits
        return x > 0.0f ? t : 3.14159274101257324f - (t - -8.74227765734758577e-8f); // pi = hi + lo, as above
    }
    const float z = x * x;
    const float t = __builtin_fmaf(x * z, asin_poly(z), x);
    return 1.57079637050628662f - (t - -4.37113882867379289e-8f);
}

// ---- atan / atan2 -----------------------------------------------------------------------------------------------------------------
SKH_LIBM_FN float atan_pos(float a) // a >= 0
{
    float y0, t;
    if (a > 2.414213562373095f) // tan(3 pi / 8)
    {
        y0 = 1.5707963267948966f;
        t = -1.0f / a;
    }
    else if (a > 0.4142135623730950f) // tan(pi / 8)
    {
        y0 = 0.7853981633974483f;
        t = (a - 1.0f) / (a + 1.0f);
    }
    else
    {
        y0 = 0.0f;
        t = a;
    }
    const float z = t * t;
    float p = 8.05374449538e-2f;
    p = __builtin_fmaf(p, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    return y0 + __builtin_fmaf(p * z, t, t);
}
SKH_LIBM_FN float atan2f_(float y, float x)
{
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
    return __ocml_atan2_f32(y, x);
#endif
    if (isnan_(x) || isnan_(y))
        return x + y;
    const float ax = fabs_(x), ay = fabs_(y);
    if (ay == 0.0f)
        return (f2u(x) >> 31) ? copysign_(3.14159274101257324f, y) : y;
    float r;
    if (ax == 0.0f)
        r = 1.5707963267948966f;
    else if (ax > 3.0e38f && ay > 3.0e38f)
        r = 0.7853981633974483f;
    else
        r = atan_pos(ay / ax);
    if (f2u(x) >> 31)
        r = 3.14159274101257324f - r;
    return copysign_(r, y);
}

// ---- exp / log / sinh / pow -------------------------------------------------------------------------------------------------------
SKH_LIBM_FN float expf_(float x)
{
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
    return __ocml_exp_f32(x);
#endif
    if (isnan_(x))
        return x;
    if (x > 88.7228394f)
        return u2f(0x7f800000u);
    if (x < -103.972084f)
        return 0.0f;
    const float kf = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(-kf, 0.693359375f, x); // ln 2 = 0.693359375 - 2.12194440e-4 (the first part has 9 significant bits: k * hi is exact)
    r = __builtin_fmaf(-kf, -2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float e = __builtin_fmaf(p * r, r, r) + 1.0f;
    return scale2(e, (int)kf);
}
SKH_LIBM_FN float logf_(float x)
{
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
    return __ocml_log_f32(x);
#endif
    if (isnan_(x))
        return x;
    if (x < 0.0f)
        return (x - x) / (x - x);
    if (x == 0.0f)
        return -u2f(0x7f800000u);
    if (x > 3.0e38f && f2u(x) == 0x7f800000u)
        return x;
    int e = 0;
    uint32_t u = f2u(x);
    if (u < 0x00800000u) // subnormal: normalise first
    {
        u = f2u(x * 8388608.0f);
        e = -23;
    }
    e += (int)(u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u); // [0.5, 1)
    if (m < 0.707106781186547524f)
    {
        e -= 1;
        m = m + m - 1.0f;
    }
    else
        m = m - 1.0f;
    const float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    const float fe = (float)e;
    float y = m * z * p;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    return __builtin_fmaf(fe, 0.693359375f, m + y);
}
SKH_LIBM_FN float sinhf_(float x)
{
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
    return __ocml_sinh_f32(x);
#endif
    const float a = fabs_(x);
    if (isnan_(x))
        return x;
    float r;
    if (a > 1.0f)
    {
        const float e = expf_(a > 89.0f ? 89.0f : a);
        r = a > 88.0f ? (a > 89.5f ? u2f(0x7f800000u) : (0.5f * expf_(0.5f * a)) * expf_(0.5f * a)) : 0.5f * e - 0.5f / e;
    }
    else
    {
        const float z = a * a;
        float p = 2.03721912945e-4f;
        p = __builtin_fmaf(p, z, 8.33028376239e-3f);
        p = __builtin_fmaf(p, z, 1.66667160211e-1f);
        r = __builtin_fmaf(p * z, a, a);
    }
    return copysign_(r, x);
}
// pow for x >= 0 (the tonemapper's gamma: Tonemappers.cu srgbGamma / pow(c, 1 / gamma)); log and exp in two floats where it matters:
// the product y * log(x) carries log's low part along, so that the result stays within a few ulp for |y log x| up to ~50
SKH_LIBM_FN float powf_(float x, float y)
{
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
    return __ocml_pow_f32(x, y);
#endif
    if (isnan_(x) || isnan_(y))
        return x + y;
    if (y == 0.0f || x == 1.0f)
        return 1.0f;
    if (x < 0.0f)
        return (x - x) / (x - x);
    if (x == 0.0f)
        return y > 0.0f ? 0.0f : u2f(0x7f800000u);
    if (f2u(x) == 0x7f800000u)
        return y > 0.0f ? x : 0.0f;
    const float lh = logf_(x);
    // one Newton correction of the logarithm in float pairs: l = lh + (x e^-lh - 1)
    const float ex = expf_(-lh);
    const float ll = __builtin_fmaf(x, ex, -1.0f);
    const float ph = y * lh;
    const float pl = __builtin_fmaf(y, lh, -ph) + y * ll;
    const float eh = expf_(ph);
    return __builtin_fmaf(eh, pl, eh); // e^(ph + pl) ~ e^ph (1 + pl)
}

} // namespace skm

// A/B only (docs/LOG.md round 5, "what the shared libm costs"): -DSKH_LIBM_NATIVE makes the DEVICE side call the ROCm device library again --
// faster where it leans on v_exp_f32 / v_log_f32 / v_sin_f32, but no longer the bits the CPU checker computes (the image tests then fail).
#if defined(SKH_LIBM_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
#    define SKH_LIBM_NATIVE_ACTIVE 1
#endif
