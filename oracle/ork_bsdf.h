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
    return f3{ r * cosf(phi), r * sinf(phi), cosTheta };
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
    const float t1 = r * cosf(phi);
    float t2 = r * sinf(phi);
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
    if (m.type == 1 || m.type == 3) // OmniPBR-like / hair (near-field tube shading)
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
    if (m.type == 2) // smooth dielectric (OmniGlass-like, thin_walled = false)
    {
        const float n1 = inside ? m.ior : 1.0f;
        const float n2 = inside ? 1.0f : m.ior;
        const float eta = n1 / n2;
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
static inline void bsdf_evaluate(const Material& m, const BsdfState& st, const f3& k1, const f3& k2, BsdfEval& out)
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
    if (m.type == 1 || m.type == 3)
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
    // glass: specular only, nothing to evaluate
}

} // namespace ork
