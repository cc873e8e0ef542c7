// MaterialDescription -> skh_material, for builds against the REAL Strelka headers (-DSKH_WITH_STRELKA_HEADERS).
//
// The reference compiles every oka::Scene::MaterialDescription {file, name, params[{type, name, value bytes}]} to MDL PTX plus an
// argument block (src/render/optix/OptixRender.cpp:1270-1433, materialmanager.cpp:524-609).  Here the block is the fixed 64-byte
// skh_material (include/strelka_hip.h) and this header is the translation -- the C++ statement of
// strelka_amd/scene_io.py::material_from_description, which the tests pin (tests/test_scene_io.py, tests/test_gltf.py):
//   default.mdl::default_material.diffuse_color          (OptixRender.cpp:1090-1097, HdStrelka/RenderPass.cpp:222-245)  -> SKH_MAT_DIFFUSE
//   OmniPBR.{diffuse_color_constant, reflection_roughness_constant, metallic_constant, diffuse_texture, normalmap_texture}
//                                                         (sceneloader/gltfloader.cpp:304-352)                            -> SKH_MAT_PBR
//   OmniGlass.{glass_color, glass_ior, frosting_roughness} (gltfloader.cpp:354-406)                                      -> SKH_MAT_GLASS
//   UsdPreviewSurface parameter sets (HdStrelka's eMaterialX descriptions, HdStrelka/Material.cpp:52-150)                -> PBR | GLASS
//   names containing "hair" (the `hair` sub-expression, materialmanager/mdlPtxCodeGen.cpp:143-155)                       -> SKH_MAT_HAIR
// The functions are templates over the description type -- anything with {file, name, params[{type, name, value}]} and the reference's
// Param::Type numbering (eFloat 0, eInt 1, eBool 2, eFloat2 3, eFloat3 4, eFloat4 5, eTexture 6: materialmanager.h:35-44) -- so that
// tests/test_integration_files.py can run them on a local look-alike of the reference's two structs and compare every case with the
// Python statement; inside the Strelka tree they are instantiated with oka::Scene::MaterialDescription itself (HipRender.cpp).
#pragma once
#ifdef SKH_WITH_STRELKA_HEADERS
#    include <strelka_hip.h>
#else
#    include "../include/strelka_hip.h"
#endif

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <string>

namespace oka
{
namespace skhmat
{
enum : uint32_t
{
    kFloat = 0,
    kInt = 1,
    kBool = 2,
    kFloat2 = 3,
    kFloat3 = 4,
    kFloat4 = 5,
    kTexture = 6
};

template <class Desc>
inline auto find(const Desc& d, const char* name) -> decltype(&d.params[0])
{
    for (const auto& p : d.params)
        if (p.name == name)
            return &p;
    return nullptr;
}
template <class Desc>
inline float scalar(const Desc& d, const char* name, float def)
{
    const auto* p = find(d, name);
    if (!p)
        return def;
    if ((uint32_t)p->type == kBool)
        return (!p->value.empty() && p->value[0]) ? 1.0f : 0.0f;
    if (p->value.size() < sizeof(float))
        return def;
    if ((uint32_t)p->type == kInt)
    {
        int v;
        memcpy(&v, p->value.data(), sizeof(v));
        return (float)v;
    }
    float v;
    memcpy(&v, p->value.data(), sizeof(v));
    return v;
}
template <class Desc>
inline void color(const Desc& d, const char* name, float out[3], float r, float g, float b)
{
    out[0] = r, out[1] = g, out[2] = b;
    const auto* p = find(d, name);
    if (p && p->value.size() >= 3 * sizeof(float))
        memcpy(out, p->value.data(), 3 * sizeof(float));
}
// the path of an eTexture parameter ("" if the material has none); the caller loads it (stbi_load(..., STBI_rgb_alpha), resolved
// against `resource/searchPath` as OptixRender.cpp:1346-1362 does) and passes the 1-based texture id back in
template <class Desc>
inline std::string texturePath(const Desc& d, const char* name)
{
    const auto* p = find(d, name);
    if (!p || (uint32_t)p->type != kTexture)
        return std::string();
    return std::string(reinterpret_cast<const char*>(p->value.data()), p->value.size());
}

template <class Desc>
inline skh_material translate(const Desc& d, uint32_t diffuseTextureId = 0, uint32_t normalTextureId = 0)
{
    skh_material m;
    memset(&m, 0, sizeof(m));
    m.base_color[0] = m.base_color[1] = m.base_color[2] = 0.8f;
    m.roughness = 0.5f, m.specular = 0.5f, m.ior = 1.5f;
    std::string low = d.name + " " + d.file;
    std::transform(low.begin(), low.end(), low.begin(), [](unsigned char c) { return (char)tolower(c); });
    const bool preview = find(d, "diffuseColor") || find(d, "useSpecularWorkflow") || find(d, "specularColor") || find(d, "clearcoat") ||
                         find(d, "emissiveColor");
    if (preview)
    {
        // UsdPreviewSurface spec defaults: diffuseColor 0.18, roughness 0.5, metallic 0, ior 1.5, opacity 1 (< 0.5 is treated as glass)
        m.type = scalar(d, "opacity", 1.0f) < 0.5f ? SKH_MAT_GLASS : SKH_MAT_PBR;
        color(d, "diffuseColor", m.base_color, 0.18f, 0.18f, 0.18f);
        m.roughness = scalar(d, "roughness", 0.5f);
        m.metallic = scalar(d, "metallic", 0.0f);
        m.ior = scalar(d, "ior", 1.5f);
    }
    else if (low.find("glass") != std::string::npos)
    {
        m.type = SKH_MAT_GLASS;
        color(d, "glass_color", m.base_color, 1.0f, 1.0f, 1.0f);
        m.roughness = scalar(d, "frosting_roughness", 0.0f);
        m.ior = scalar(d, "glass_ior", 1.491f); // OmniGlass.mdl default
    }
    else if (low.find("pbr") != std::string::npos)
    {
        m.type = SKH_MAT_PBR;
        color(d, "diffuse_color_constant", m.base_color, 0.2f, 0.2f, 0.2f); // OmniPBR.mdl default
        m.roughness = scalar(d, "reflection_roughness_constant", 0.5f);
        m.metallic = scalar(d, "metallic_constant", 0.0f);
        m.base_color_texture = diffuseTextureId;
        m.normal_texture = normalTextureId;
    }
    else if (low.find("hair") != std::string::npos)
    {
        // df::chiang_hair_bsdf: colour -> absorption by Chiang et al. 2016 eq. 9 when no absorption_coefficient is given
        m.type = SKH_MAT_HAIR;
        const float rn = scalar(d, "roughness_azimuthal", scalar(d, "roughness", 0.3f));
        float sig[3];
        if (find(d, "absorption_coefficient"))
            color(d, "absorption_coefficient", sig, 0.0f, 0.0f, 0.0f);
        else
        {
            float c[3];
            color(d, find(d, "diffuse_color") ? "diffuse_color" : "color", c, 0.35f, 0.2f, 0.1f);
            const double b = rn, dn = 5.969 - 0.215 * b + 2.532 * b * b - 10.73 * b * b * b + 5.574 * b * b * b * b + 0.245 * b * b * b * b * b;
            for (int k = 0; k < 3; ++k)
            {
                const double l = std::log(std::min(1.0, std::max(1e-4, (double)c[k]))) / dn;
                sig[k] = (float)(l * l);
            }
        }
        color(d, "diffuse_reflection_tint", m.base_color, 1.0f, 1.0f, 1.0f);
        m.roughness = scalar(d, "roughness_R", scalar(d, "roughness", 0.3f));
        m.metallic = scalar(d, "roughness_TT", 0.0f);
        m.specular = scalar(d, "roughness_TRT", 0.0f);
        m.ior = scalar(d, "ior", 1.55f);
        m.reserved[0] = sig[0], m.reserved[1] = sig[1], m.reserved[2] = sig[2], m.reserved[3] = rn;
        m.reserved[4] = scalar(d, "cuticle_angle", 0.035f);
        m.reserved[5] = scalar(d, "diffuse_reflection_weight", 0.0f);
    }
    else
    {
        m.type = SKH_MAT_DIFFUSE;
        color(d, "diffuse_color", m.base_color, 0.8f, 0.8f, 0.8f);
    }
    return m;
}
} // namespace skhmat
} // namespace oka
