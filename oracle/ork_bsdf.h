// ORACLE -- TEST INFRASTRUCTURE ONLY (see ork_math.h).
//
// ork_bsdf.h: the "MDL-equivalent" BSDF set.  In the reference the BSDF is PTX
// generated at run time by the closed NVIDIA MDL SDK (mdlcode_init / mdlcode_sample / mdlcode_evaluate:
// src/render/optix/OptixRender_radiance_closest_hit.cu:31-33) => PARITY UNPINNED for the arithmetic.
// What IS restated from the reference is the call protocol (closest_hit.cu:477-605):
//   state { position, normal, geom_normal, ... };  sample(k1 = -ray_dir, xi = float4, ior1/ior2)
//     -> { k2, bsdf_over_pdf, pdf, event_type };   evaluate(k1, k2) -> { bsdf_diffuse, bsdf_glossy, pdf }
//   bsdf_diffuse / bsdf_glossy INCLUDE the cosine term; specular events report pdf = 0.
// and the only in-source BSDF, the Metal backend's Lambert (src/render/metal/shaders/pathtrace.metal:164-201),
// which defines the diffuse-only configuration: bsdf*cos = albedo * (n.k2) / pi, pdf = (n.k2) / pi,
// sample = cosine-weighted, bsdf_over_pdf = albedo.  (pathtrace.metal:198 sets pdf = a/pi from the z of the
// unit-sphere point rather than n.k2 -- a reference bug; this build uses (n.k2)/pi and normalises k2.)
//
// Side handling follows MDL's libbsdf convention: if k1 is on the back side of the geometric normal, both
// normals are flipped for the BSDF (two-sided), while the renderer's NEE test keeps using state.normal.
#pragma once
#include "ork_trace.h"

namespace ork
{

// mi::neuraylib::Bsdf_event_type bits
enum
{
    EV_ABSORB = 0,
    EV_DIFFUSE = 1,
    EV_GLOSSY = 2,
    EV_SPECULAR = 4,
    EV_REFLECTION = 8,
    EV_TRANSMISSION = 16
};

struct Material // == skh_material, 64 B
{
    uint32_t type;
    float base_color[3];
    float roughness;
    float metallic;
    float specular;
    float ior;
    uint32_t base_color_texture; // 1-based index into the texture list, 0 = none (MDL texture ids: 0 is the invalid texture)
    uint32_t normal_texture;
    float reserved[6];
};
static_assert(sizeof(Material) == 64, "Material layout");

struct BsdfState
{
    f3 normal; // state.normal
    f3 geom_normal; // state.geom_normal
    f3 tangent_u; // state.tangent_u[0]: the curve tangent for hair (closest_hit.cu:436-437,446); unused by the surface BSDFs
};
struct BsdfSample
{
    f3 k2;
    f3 bsdf_over_pdf;
    float pdf;
    int event_type;
};
struct BsdfEval
{
    f3 bsdf_diffuse;
    f3 bsdf_glossy;
    float pdf;
};

static inline f3 cosine_hemisphere(float u1, float u2, float& cosTheta)
{
    const float r = sqrtf(u1);
    const float phi = 2.0f * kPi * u2;
    cosTheta = sqrtf(fmaxf(0.0f, 1.0f - u1));
    return f3{ r * skm::cosf_(phi), r * skm::sinf_(phi), cosTheta };
}
static inline float lum3(const f3& c)
{
    return 0.299f * c.x + 0.587f * c.y + 0.114f * c.z;
}
static inline f3 schlick3(const f3& f0, float c)
{
    const float m = clampf(1.0f - c, 0.0f, 1.0f);
    const float m2 = m * m;
    const float m5 = m2 * m2 * m;
    return f0 + (mk3(1.0f) - f0) * m5;
}
// GGX helpers in the local frame (z = normal)
static inline float ggx_D(float alpha, float nh)
{
    const float a2 = alpha * alpha;
    const float d = nh * nh * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * d * d);
}
static inline float ggx_lambda(float alpha, float cosT)
{
    const float c2 = cosT * cosT;
    const float t2 = fmaxf(0.0f, 1.0f - c2) / fmaxf(c2, 1e-20f);
    return 0.5f * (sqrtf(1.0f + alpha * alpha * t2) - 1.0f);
}
// Heitz 2018, "Sampling the GGX distribution of visible normals" (isotropic)
static inline f3 ggx_sample_vndf(const f3& Ve, float alpha, float u1, float u2)
{
    const f3 Vh = normalize(f3{ alpha * Ve.x, alpha * Ve.y, Ve.z });
    const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    const f3 T1 = lensq > 0.0f ? f3{ -Vh.y, Vh.x, 0.0f } * (1.0f / sqrtf(lensq)) : f3{ 1.0f, 0.0f, 0.0f };
    const f3 T2 = cross(Vh, T1);
    const float r = sqrtf(u1);
    const float phi = 2.0f * kPi * u2;
    const float t1 = r * skm::cosf_(phi);
    float t2 = r * skm::sinf_(phi);
    const float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1)) + s * t2;
    const f3 Nh = t1 * T1 + t2 * T2 + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(f3{ alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z) });
}

struct PbrTerms
{
    f3 diffuse_albedo;
    f3 f0;
    float alpha;
    float p_spec;
};
static inline PbrTerms pbr_terms(const Material& m)
{
    PbrTerms t;
    const f3 base{ m.base_color[0], m.base_color[1], m.base_color[2] };
    const float metallic = clampf(m.metallic, 0.0f, 1.0f);
    t.diffuse_albedo = base * (1.0f - metallic);
    const float d = 0.08f * m.specular;
    t.f0 = mk3(d) + (base - mk3(d)) * metallic;
    const float r = fmaxf(m.roughness, 0.05f);
    t.alpha = r * r;
    t.p_spec = 0.5f + 0.5f * metallic;
    return t;
}
// local-frame evaluation shared by sample and evaluate
static inline void pbr_eval_local(const PbrTerms& t, const f3& wo, const f3& wi, f3& fd, f3& fs, float& pdf)
{
    fd = mk3(0.0f);
    fs = mk3(0.0f);
    pdf = 0.0f;
    if (wo.z <= 0.0f || wi.z <= 0.0f)
        return;
    const f3 h = normalize(wo + wi);
    const float oh = fmaxf(dot(wo, h), 0.0f);
    const f3 F = schlick3(t.f0, oh);
    const float D = ggx_D(t.alpha, h.z);
    const float lo = ggx_lambda(t.alpha, wo.z), li = ggx_lambda(t.alpha, wi.z);
    const float G2 = 1.0f / (1.0f + lo + li);
    const float G1 = 1.0f / (1.0f + lo);
    fs = F * (D * G2 / (4.0f * wo.z)); // f * cos(wi)
    const f3 Fo = schlick3(t.f0, wo.z);
    fd = t.diffuse_albedo * (mk3(1.0f) - Fo) * (wi.z / kPi);
    const float pdf_s = G1 * D / (4.0f * wo.z);
    const float pdf_d = wi.z / kPi;
    pdf = t.p_spec * pdf_s + (1.0f - t.p_spec) * pdf_d;
}

static inline float fresnel_dielectric(float cosi, float eta /* n1/n2 */, float& cost)
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
// Rough dielectric (OmniGlass with frosting_roughness > 0: gltfloader.cpp:354-406 sets it from the glTF roughnessFactor).
// Published model: Walter et al. 2007, "Microfacet Models for Refraction through Rough Surfaces", GGX distribution with
// alpha = frosting_roughness^2, visible-normal sampling (Heitz 2018).  Local frame: z = N on the side of k1, so wo.z > 0.
// eta = n1 / n2 (incident / transmitted), as in fresnel_dielectric.  Weights: reflection G2/G1, transmission tint * G2/G1
// (no radiance scaling, like the smooth branch); events are GLOSSY, so next-event estimation applies (closest_hit.cu:538).
// ------------------------------------------------------------------------------------------------------------
#define ORK_GLASS_SMOOTH_BELOW 1e-3f // frosting_roughness below this = the specular (delta) branch
static inline float ggx_G1(float alpha, float cosT)
{
    return 1.0f / (1.0f + ggx_lambda(alpha, fabsf(cosT)));
}
// f * |cos(wi)| and pdf of the rough dielectric for a given pair of directions (wo.z > 0; wi on either side)
static inline void rough_glass_eval_local(float alpha, float eta, const f3& tint, const f3& wo, const f3& wi, f3& f_cos, float& pdf)
{
    f_cos = mk3(0.0f);
    pdf = 0.0f;
    if (wo.z <= 0.0f || wi.z == 0.0f)
        return;
    const bool reflect = wi.z > 0.0f;
    // half vector: reflection h = wo + wi; refraction h = -(eta * wo + wi) (points to the incident side after the flip below)
    f3 h = reflect ? wo + wi : (wo * eta + wi);
    const float hl = sqrtf(dot(h, h));
    if (!(hl > 0.0f))
        return;
    h = h * (1.0f / hl);
    if (h.z < 0.0f)
        h = -h;
    const float oh = dot(wo, h), ih = dot(wi, h);
    if (oh <= 0.0f || (reflect ? ih <= 0.0f : ih >= 0.0f))
        return; // the micro-normal must face wo, and wi must be on the matching side of it
    float cost;
    const float F = fresnel_dielectric(fminf(oh, 1.0f), eta, cost);
    const float D = ggx_D(alpha, h.z);
    const float lo = ggx_lambda(alpha, wo.z), li = ggx_lambda(alpha, fabsf(wi.z));
    const float G1o = 1.0f / (1.0f + lo);
    const float G2 = 1.0f / (1.0f + lo + li);
    const float pdf_h = G1o * D * oh / wo.z; // visible-normal density of h
    if (reflect)
    {
        const float jac = 1.0f / (4.0f * oh);
        pdf = F * pdf_h * jac;
        f_cos = mk3(F * D * G2 / (4.0f * wo.z));
    }
    else
    {
        const float denom = eta * oh + ih; // (wi.h < 0)
        const float jac = fabsf(ih) / (denom * denom);
        pdf = (1.0f - F) * pdf_h * jac;
        f_cos = tint * ((1.0f - F) * D * G2 * oh * jac / wo.z);
    }
}
static inline bool rough_glass_sample_local(float alpha, float eta, const f3& tint, const f3& wo, float u0, float u1, float u2, f3& wi,
                                            f3& weight, float& pdf, bool& transmitted)
{
    const f3 h = ggx_sample_vndf(wo, alpha, u0, u1);
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
    f3 f_cos;
    rough_glass_eval_local(alpha, eta, tint, wo, wi, f_cos, pdf);
    if (!(pdf > 0.0f))
        return false;
    weight = f_cos / pdf;
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// Hair: df::chiang_hair_bsdf, the MDL distribution function the reference compiles for hair materials (the `hair`
// sub-expression: mdlPtxCodeGen.cpp:143-155).  Closed arithmetic (MDL SDK) => restated from the published model: Chiang, Bitterli,
// Tappan, Burley 2016, "A Practical and Controllable Hair and Fur Model for Production Path Tracing", in the formulation
// of pbrt-v3 (hair.cpp), which MDL's libbsdf follows: lobes R, TT, TRT + residual, longitudinal M_p (von Mises-like with
// variance v), azimuthal N_p (trimmed logistic of scale s around the ideal-specular exit azimuth), attenuations A_p from one
// Fresnel term and the absorption through the fibre.
//
// Parameters = the arguments of df::chiang_hair_bsdf, carried in skh_material (type 3):
//   base_color       diffuse_reflection_tint        reserved[0..2]  absorption_coefficient (per unit fibre diameter)
//   roughness        roughness_R.x  (longitudinal)  reserved[3]     roughness_*.y (azimuthal, shared by the lobes)
//   metallic         roughness_TT.x (<= 0: v_TT = v_R / 4, the paper's default)
//   specular         roughness_TRT.x (<= 0: v_TRT = 4 v_R)            reserved[4]     cuticle_angle (radians)
//   ior              ior                            reserved[5]     diffuse_reflection_weight
// Frame: x = state.tangent_u (along the fibre), z = state.normal made orthogonal to it, y = z x x.  The offset across the fibre
// h = 2 * state.text_coords[0].y - 1, and the reference sets text_coords[0] = (0.5, 0.5, 0.5) for curve hits
// (closest_hit.cu:445): h = 0, every ray is shaded as if it went through the fibre's axis.  Kept, and visible below.
// All events are reported as GLOSSY | REFLECTION (a fibre has no inside; TT leaves through the far side of the tube).
// ------------------------------------------------------------------------------------------------------------
#define ORK_HAIR_TEXCOORD_Y 0.5f // state.text_coords[0].y of a curve hit (closest_hit.cu:445)
static inline float sqrf(float x)
{
    return x * x;
}
static inline float safe_sqrtf(float x)
{
    return sqrtf(fmaxf(0.0f, x));
}
static inline float safe_asinf(float x)
{
    return skm::asinf_(clampf(x, -1.0f, 1.0f));
}
static inline float hair_pow20(float x)
{
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    return x16 * x4;
}
static inline float hair_pow22(float x)
{
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    return (x16 * x4) * x2;
}
// modified Bessel function I0, the ten-term series of pbrt-v3 (hair.h), Horner form in y = x^2: sum_i y^i / (4^i (i!)^2)
static inline float hair_I0(float x)
{
    const float y = x * x;
    float v = 1.0f / 34519618525593600.0f; // 4^9 (9!)^2
    v = v * y + 1.0f / 106542032486400.0f; // 4^8 (8!)^2
    v = v * y + 1.0f / 416179814400.0f; // 4^7 (7!)^2
    v = v * y + 1.0f / 2123366400.0f; // 4^6 (6!)^2
    v = v * y + 1.0f / 14745600.0f;
    v = v * y + 1.0f / 147456.0f;
    v = v * y + 1.0f / 2304.0f;
    v = v * y + 1.0f / 64.0f;
    v = v * y + 0.25f;
    v = v * y + 1.0f;
    return v;
}
static inline float hair_logI0(float x)
{
    if (x > 12.0f)
        return x + 0.5f * ((-skm::logf_(2.0f * kPi) + skm::logf_(1.0f / x)) + 1.0f / (8.0f * x));
    return skm::logf_(hair_I0(x));
}
static inline float hair_Mp(float cosThetaI, float cosThetaO, float sinThetaI, float sinThetaO, float v)
{
    const float a = cosThetaI * cosThetaO / v;
    const float b = sinThetaI * sinThetaO / v;
    return v <= 0.1f ? skm::expf_((((hair_logI0(a) - b) - 1.0f / v) + 0.6931f) + skm::logf_(1.0f / (2.0f * v))) :
                       (skm::expf_(-b) * hair_I0(a)) / (skm::sinhf_(1.0f / v) * 2.0f * v);
}
static inline float hair_logistic(float x, float s)
{
    x = fabsf(x);
    const float e = skm::expf_(-x / s);
    return e / (s * sqrf(1.0f + e));
}
static inline float hair_logistic_cdf(float x, float s)
{
    return 1.0f / (1.0f + skm::expf_(-x / s));
}
static inline float hair_trimmed_logistic(float x, float s, float a, float b)
{
    return hair_logistic(x, s) / (hair_logistic_cdf(b, s) - hair_logistic_cdf(a, s));
}
static inline float hair_sample_trimmed_logistic(float u, float s, float a, float b)
{
    const float k = hair_logistic_cdf(b, s) - hair_logistic_cdf(a, s);
    const float x = -s * skm::logf_(1.0f / (u * k + hair_logistic_cdf(a, s)) - 1.0f);
    return clampf(x, a, b);
}
static inline float hair_Phi(int p, float gammaO, float gammaT)
{
    return (2.0f * (float)p * gammaT - 2.0f * gammaO) + (float)p * kPi;
}
static inline float hair_Np(float phi, int p, float s, float gammaO, float gammaT)
{
    float dphi = phi - hair_Phi(p, gammaO, gammaT);
    while (dphi > kPi)
        dphi -= 2.0f * kPi;
    while (dphi < -kPi)
        dphi += 2.0f * kPi;
    return hair_trimmed_logistic(dphi, s, -kPi, kPi);
}
static inline float hair_variance(float roughness) // longitudinal roughness -> variance of M_p (Chiang et al. 2016, eq. 7)
{
    const float r = fmaxf(roughness, 0.02f);
    return sqrf((0.726f * r + 0.812f * (r * r)) + 3.7f * hair_pow20(r));
}
struct HairTerms
{
    float h, eta, s, diffuse_w;
    f3 sigma_a, tint;
    float v[4];
    float sin2k[3], cos2k[3];
};
static inline HairTerms hair_terms(const Material& m)
{
    HairTerms t;
    t.h = 2.0f * ORK_HAIR_TEXCOORD_Y - 1.0f;
    t.eta = m.ior > 1.0f ? m.ior : 1.55f;
    t.sigma_a = f3{ fmaxf(m.reserved[0], 0.0f), fmaxf(m.reserved[1], 0.0f), fmaxf(m.reserved[2], 0.0f) };
    t.tint = f3{ m.base_color[0], m.base_color[1], m.base_color[2] };
    t.diffuse_w = clampf(m.reserved[5], 0.0f, 1.0f);
    t.v[0] = hair_variance(m.roughness);
    t.v[1] = m.metallic > 0.0f ? hair_variance(m.metallic) : 0.25f * t.v[0];
    t.v[2] = m.specular > 0.0f ? hair_variance(m.specular) : 4.0f * t.v[0];
    t.v[3] = t.v[2];
    const float bn = fmaxf(m.reserved[3] > 0.0f ? m.reserved[3] : m.roughness, 0.02f);
    t.s = 0.626657069f * ((0.265f * bn + 1.194f * (bn * bn)) + 5.372f * hair_pow22(bn)); // sqrt(pi / 8) * (...)
    t.sin2k[0] = skm::sinf_(m.reserved[4]);
    t.cos2k[0] = safe_sqrtf(1.0f - sqrf(t.sin2k[0]));
    for (int i = 1; i < 3; ++i)
    {
        t.sin2k[i] = 2.0f * t.cos2k[i - 1] * t.sin2k[i - 1];
        t.cos2k[i] = sqrf(t.cos2k[i - 1]) - sqrf(t.sin2k[i - 1]);
    }
    return t;
}
// attenuations A_p (colour) and their normalised luminances (lobe selection probabilities)
static inline void hair_Ap(const HairTerms& t, float cosThetaO, const f3& T, f3 ap[4], float apPdf[4])
{
    const float cosGammaO = safe_sqrtf(1.0f - t.h * t.h);
    const float cosTheta = cosThetaO * cosGammaO;
    float cost;
    const float f = fresnel_dielectric(clampf(cosTheta, 0.0f, 1.0f), 1.0f / t.eta, cost);
    ap[0] = mk3(f);
    ap[1] = T * sqrf(1.0f - f);
    ap[2] = ap[1] * T * f;
    ap[3] = (ap[2] * f) * T / (mk3(1.0f) - T * f);
    float y[4], sum = 0.0f;
    for (int p = 0; p < 4; ++p)
    {
        y[p] = lum3(ap[p]);
        sum += y[p];
    }
    for (int p = 0; p < 4; ++p)
        apPdf[p] = sum > 0.0f ? y[p] / sum : 0.25f;
}
struct HairGeom
{
    float sinThetaO, cosThetaO, phiO, gammaO, gammaT;
    f3 T;
};
static inline HairGeom hair_geom(const HairTerms& t, const f3& wo)
{
    HairGeom g;
    g.sinThetaO = clampf(wo.x, -1.0f, 1.0f);
    g.cosThetaO = safe_sqrtf(1.0f - sqrf(g.sinThetaO));
    g.phiO = skm::atan2f_(wo.z, wo.y);
    const float sinThetaT = g.sinThetaO / t.eta;
    const float cosThetaT = safe_sqrtf(1.0f - sqrf(sinThetaT));
    const float etap = sqrtf(t.eta * t.eta - sqrf(g.sinThetaO)) / fmaxf(g.cosThetaO, 1e-6f);
    const float sinGammaT = t.h / etap;
    const float cosGammaT = safe_sqrtf(1.0f - sqrf(sinGammaT));
    g.gammaT = safe_asinf(sinGammaT);
    g.gammaO = safe_asinf(t.h);
    const float l = 2.0f * cosGammaT / fmaxf(cosThetaT, 1e-6f);
    g.T = f3{ skm::expf_(-t.sigma_a.x * l), skm::expf_(-t.sigma_a.y * l), skm::expf_(-t.sigma_a.z * l) };
    return g;
}
static inline void hair_tilt(const HairTerms& t, const HairGeom& g, int p, float& sinThetaOp, float& cosThetaOp)
{
    // the cuticle scales tilt the lobes: R by 2 alpha towards the root, TT by -alpha, TRT by -4 alpha (pbrt-v3 hair.cpp)
    if (p == 0)
    {
        sinThetaOp = g.sinThetaO * t.cos2k[1] - g.cosThetaO * t.sin2k[1];
        cosThetaOp = g.cosThetaO * t.cos2k[1] + g.sinThetaO * t.sin2k[1];
    }
    else if (p == 1)
    {
        sinThetaOp = g.sinThetaO * t.cos2k[0] + g.cosThetaO * t.sin2k[0];
        cosThetaOp = g.cosThetaO * t.cos2k[0] - g.sinThetaO * t.sin2k[0];
    }
    else if (p == 2)
    {
        sinThetaOp = g.sinThetaO * t.cos2k[2] + g.cosThetaO * t.sin2k[2];
        cosThetaOp = g.cosThetaO * t.cos2k[2] - g.sinThetaO * t.sin2k[2];
    }
    else
    {
        sinThetaOp = g.sinThetaO;
        cosThetaOp = g.cosThetaO;
    }
    cosThetaOp = fabsf(cosThetaOp);
}
// f * |cos| (colour) and pdf of the fibre lobes for local directions wo, wi (hair frame: x along the fibre)
static inline void hair_eval_local(const HairTerms& t, const f3& wo, const f3& wi, f3& f_cos, float& pdf)
{
    const HairGeom g = hair_geom(t, wo);
    const float sinThetaI = clampf(wi.x, -1.0f, 1.0f);
    const float cosThetaI = safe_sqrtf(1.0f - sqrf(sinThetaI));
    const float phi = skm::atan2f_(wi.z, wi.y) - g.phiO;
    f3 ap[4];
    float apPdf[4];
    hair_Ap(t, g.cosThetaO, g.T, ap, apPdf);
    f_cos = mk3(0.0f);
    pdf = 0.0f;
    for (int p = 0; p < 3; ++p)
    {
        float so, co;
        hair_tilt(t, g, p, so, co);
        const float mn = hair_Mp(cosThetaI, co, sinThetaI, so, t.v[p]) * hair_Np(phi, p, t.s, g.gammaO, g.gammaT);
        f_cos += ap[p] * mn;
        pdf += apPdf[p] * mn;
    }
    const float mr = hair_Mp(cosThetaI, g.cosThetaO, sinThetaI, g.sinThetaO, t.v[3]) * (1.0f / (2.0f * kPi));
    f_cos += ap[3] * mr;
    pdf += apPdf[3] * mr;
}
static inline f3 hair_sample_local(const HairTerms& t, const f3& wo, float u0, float u1, float u2, float u3)
{
    const HairGeom g = hair_geom(t, wo);
    f3 ap[4];
    float apPdf[4];
    hair_Ap(t, g.cosThetaO, g.T, ap, apPdf);
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
    const float ua = fmaxf(u0, 1e-5f);
    const float cosTheta = 1.0f + t.v[p] * skm::logf_(ua + (1.0f - ua) * skm::expf_(-2.0f / t.v[p]));
    const float sinTheta = safe_sqrtf(1.0f - sqrf(cosTheta));
    const float cosPhi = skm::cosf_(2.0f * kPi * u1);
    const float sinThetaI = -cosTheta * so + sinTheta * cosPhi * co;
    const float cosThetaI = safe_sqrtf(1.0f - sqrf(sinThetaI));
    const float dphi = p < 3 ? hair_Phi(p, g.gammaO, g.gammaT) + hair_sample_trimmed_logistic(u3, t.s, -kPi, kPi) : 2.0f * kPi * u3;
    const float phiI = g.phiO + dphi;
    return f3{ sinThetaI, cosThetaI * skm::cosf_(phiI), cosThetaI * skm::sinf_(phiI) };
}
// hair frame from the MDL state: x = tangent_u, z = normal orthogonalised against it, y = z x x
static inline bool hair_frame(const BsdfState& st, f3& X, f3& Y, f3& Z)
{
    const float tl = dot(st.tangent_u, st.tangent_u);
    if (!(tl > 0.0f))
        return false;
    X = st.tangent_u * (1.0f / sqrtf(tl));
    f3 z = st.normal - X * dot(st.normal, X);
    const float zl = dot(z, z);
    if (!(zl > 1e-12f))
        return false;
    Z = z * (1.0f / sqrtf(zl));
    Y = cross(Z, X);
    return true;
}

// mdlcode_sample equivalent.  `inside` selects ior1/ior2 exactly as closest_hit.cu:496-498 does.
static inline void bsdf_sample(const Material& m, const BsdfState& st, const f3& k1, const float xi[4], bool inside,
                               BsdfSample& out)
{
    f3 N = st.normal, Ng = st.geom_normal;
    if (dot(Ng, k1) < 0.0f)
    {
        N = -N;
        Ng = -Ng;
    }
    f3 b1, b2;
    onb_from_z(N, b1, b2);
    const f3 wo{ dot(k1, b1), dot(k1, b2), dot(k1, N) };
    out.k2 = mk3(0.0f);
    out.bsdf_over_pdf = mk3(0.0f);
    out.pdf = 0.0f;
    out.event_type = EV_ABSORB;
    const f3 base{ m.base_color[0], m.base_color[1], m.base_color[2] };

    if (m.type == 0) // diffuse
    {
        float cosT;
        const f3 w = cosine_hemisphere(xi[0], xi[1], cosT);
        const f3 k2 = normalize(w.x * b1 + w.y * b2 + w.z * N);
        if (cosT <= 0.0f || dot(k2, Ng) <= 0.0f)
            return;
        out.k2 = k2;
        out.pdf = cosT / kPi;
        out.bsdf_over_pdf = base;
        out.event_type = EV_DIFFUSE | EV_REFLECTION;
        return;
    }
    if (m.type == 3) // df::chiang_hair_bsdf
    {
        f3 X, Y, Z;
        if (!hair_frame(st, X, Y, Z))
            return;
        const HairTerms t = hair_terms(m);
        const f3 ho{ dot(k1, X), dot(k1, Y), dot(k1, Z) };
        float u2 = xi[2];
        if (u2 < t.diffuse_w)
        {
            // diffuse_reflection_weight: a Lambert lobe about the surface normal, tinted
            float cosT;
            const f3 w = cosine_hemisphere(xi[0], xi[1], cosT);
            const f3 k2 = normalize(w.x * b1 + w.y * b2 + w.z * N);
            if (cosT <= 0.0f)
                return;
            const f3 hi{ dot(k2, X), dot(k2, Y), dot(k2, Z) };
            f3 fh;
            float ph;
            hair_eval_local(t, ho, hi, fh, ph);
            const float pd = cosT / kPi;
            const float pdf = t.diffuse_w * pd + (1.0f - t.diffuse_w) * ph;
            out.k2 = k2;
            out.pdf = pdf;
            out.bsdf_over_pdf = (t.tint * (t.diffuse_w * pd) + fh * (1.0f - t.diffuse_w)) / pdf;
            out.event_type = EV_DIFFUSE | EV_REFLECTION;
            return;
        }
        u2 = (u2 - t.diffuse_w) / (1.0f - t.diffuse_w);
        const f3 hi = hair_sample_local(t, ho, xi[0], xi[1], u2, xi[3]);
        f3 fh;
        float ph;
        hair_eval_local(t, ho, hi, fh, ph);
        const f3 k2 = normalize(hi.x * X + hi.y * Y + hi.z * Z);
        const float cosN = dot(k2, N);
        const float pd = cosN > 0.0f ? cosN / kPi : 0.0f;
        const float pdf = t.diffuse_w * pd + (1.0f - t.diffuse_w) * ph;
        if (!(pdf > 0.0f) || !(ph > 0.0f))
            return;
        out.k2 = k2;
        out.pdf = pdf;
        out.bsdf_over_pdf = (t.tint * (t.diffuse_w * pd) + fh * (1.0f - t.diffuse_w)) / pdf;
        out.event_type = EV_GLOSSY | EV_REFLECTION;
        return;
    }
    if (m.type == 1) // OmniPBR-like
    {
        if (wo.z <= 0.0f)
            return;
        const PbrTerms t = pbr_terms(m);
        f3 wi;
        int ev;
        if (xi[2] < t.p_spec)
        {
            const f3 h = ggx_sample_vndf(wo, t.alpha, xi[0], xi[1]);
            wi = h * (2.0f * dot(wo, h)) - wo;
            ev = EV_GLOSSY | EV_REFLECTION;
        }
        else
        {
            float cosT;
            wi = cosine_hemisphere(xi[0], xi[1], cosT);
            ev = EV_DIFFUSE | EV_REFLECTION;
        }
        if (wi.z <= 0.0f)
            return;
        const f3 k2 = normalize(wi.x * b1 + wi.y * b2 + wi.z * N);
        if (dot(k2, Ng) <= 0.0f)
            return;
        f3 fd, fs;
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
    if (m.type == 2) // dielectric (OmniGlass-like, thin_walled = false): smooth, or frosted when frosting_roughness > 0
    {
        const float n1 = inside ? m.ior : 1.0f;
        const float n2 = inside ? 1.0f : m.ior;
        const float eta = n1 / n2;
        if (m.roughness >= ORK_GLASS_SMOOTH_BELOW)
        {
            if (wo.z <= 0.0f)
                return;
            const float alpha = fmaxf(m.roughness * m.roughness, 1e-4f);
            f3 wi, weight;
            float pdf;
            bool transmitted;
            if (!rough_glass_sample_local(alpha, eta, base, wo, xi[0], xi[1], xi[2], wi, weight, pdf, transmitted))
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
        if (xi[2] < F)
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

// mdlcode_evaluate equivalent
static inline void bsdf_evaluate(const Material& m, const BsdfState& st, const f3& k1, const f3& k2, bool inside, BsdfEval& out)
{
    f3 N = st.normal, Ng = st.geom_normal;
    if (dot(Ng, k1) < 0.0f)
    {
        N = -N;
        Ng = -Ng;
    }
    out.bsdf_diffuse = mk3(0.0f);
    out.bsdf_glossy = mk3(0.0f);
    out.pdf = 0.0f;
    const f3 base{ m.base_color[0], m.base_color[1], m.base_color[2] };
    if (m.type == 0)
    {
        const float nk2 = dot(N, k2);
        if (nk2 <= 0.0f || dot(Ng, k2) <= 0.0f)
            return;
        out.bsdf_diffuse = base * (nk2 / kPi);
        out.pdf = nk2 / kPi;
        return;
    }
    if (m.type == 3)
    {
        f3 X, Y, Z;
        if (!hair_frame(st, X, Y, Z))
            return;
        const HairTerms t = hair_terms(m);
        const f3 ho{ dot(k1, X), dot(k1, Y), dot(k1, Z) };
        const f3 hi{ dot(k2, X), dot(k2, Y), dot(k2, Z) };
        f3 fh;
        float ph;
        hair_eval_local(t, ho, hi, fh, ph);
        const float cosN = dot(k2, N);
        const float pd = cosN > 0.0f ? cosN / kPi : 0.0f;
        out.bsdf_glossy = fh * (1.0f - t.diffuse_w);
        out.bsdf_diffuse = t.tint * (t.diffuse_w * pd);
        out.pdf = t.diffuse_w * pd + (1.0f - t.diffuse_w) * ph;
        return;
    }
    if (m.type == 1)
    {
        f3 b1, b2;
        onb_from_z(N, b1, b2);
        const f3 wo{ dot(k1, b1), dot(k1, b2), dot(k1, N) };
        const f3 wi{ dot(k2, b1), dot(k2, b2), dot(k2, N) };
        if (dot(Ng, k2) <= 0.0f)
            return;
        const PbrTerms t = pbr_terms(m);
        pbr_eval_local(t, wo, wi, out.bsdf_diffuse, out.bsdf_glossy, out.pdf);
        return;
    }
    if (m.type == 2 && m.roughness >= ORK_GLASS_SMOOTH_BELOW)
    {
        f3 b1, b2;
        onb_from_z(N, b1, b2);
        const f3 wo{ dot(k1, b1), dot(k1, b2), dot(k1, N) };
        const f3 wi{ dot(k2, b1), dot(k2, b2), dot(k2, N) };
        const float n1 = inside ? m.ior : 1.0f;
        const float n2 = inside ? 1.0f : m.ior;
        rough_glass_eval_local(fmaxf(m.roughness * m.roughness, 1e-4f), n1 / n2, base, wo, wi, out.bsdf_glossy, out.pdf);
        return;
    }
    // smooth glass: specular only, nothing to evaluate
}

} // namespace ork
