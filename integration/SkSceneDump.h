// skhDumpScene: writes what HipRender::render() uploads on its first frame -- oka::Scene's flat arrays -- as a ".skscene" file
// (format: strelka_amd/scene_io.py).  This is the exporter a Strelka build runs once after its USD / glTF bake so that the real
// Kitchen_set / Einar bakes can reach a machine that has no OpenUSD: integration/strelka_hip.patch calls it from
// HdStrelkaRenderPass::_Execute right after _BakeMeshes (src/HdStrelka/RenderPass.cpp:362-366) and after GltfLoader::loadGltf
// (src/app/main.cpp) when the environment variable STRELKA_DUMP_SKSCENE names a file.
//
// Header-only, both header sets (see HipRender.h): with the real headers the materials travel as their original descriptions (MDSC: JSON
// of {file, name, params[{name, type, value}]}; strelka_amd/scene_io.py::materials_from_descriptions maps them), with the mirror as the
// skh_material blocks it already holds (MATL) -- byte-identical to oka::Scene::saveDump there (strelka_amd/host/host_test.cpp checks).
#pragma once
#ifdef SKH_WITH_STRELKA_HEADERS
#    include <scene/scene.h>
#    include <strelka_hip.h>
#else
#    include "../strelka_amd/host/oka_mirror.h"
#endif

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace oka
{
namespace skhdump
{
inline void putSection(FILE* f, const char tag[4], uint32_t elemSize, uint64_t count, const void* data)
{
    fwrite(tag, 1, 4, f);
    fwrite(&elemSize, 4, 1, f);
    fwrite(&count, 8, 1, f);
    const uint64_t bytes = (uint64_t)elemSize * count;
    if (bytes)
        fwrite(data, 1, bytes, f);
    static const char zero[8] = { 0 };
    fwrite(zero, 1, (size_t)((8 - bytes % 8) % 8), f);
}
struct CameraRecord
{
    float view[16]; // world -> view, row-major
    float fov, znear, zfar;
    uint32_t pad[5];
};
static_assert(sizeof(CameraRecord) == 96, "camera record");

#ifdef SKH_WITH_STRELKA_HEADERS
inline std::string jsonEscape(const std::string& s)
{
    std::string o;
    for (char ch : s)
    {
        if (ch == '"' || ch == '\\')
            o += '\\';
        if ((unsigned char)ch >= 0x20)
            o += ch;
    }
    return o;
}
// [{"file", "name", "params": [{"name", "type", "value"}]}]: floats as numbers / arrays, ints and bools as numbers, textures as their path
inline std::string materialsToJson(const std::vector<Scene::MaterialDescription>& mats)
{
    using P = MaterialManager::Param;
    std::string j = "[";
    for (size_t m = 0; m < mats.size(); ++m)
    {
        j += (m ? ",{" : "{");
        j += "\"file\":\"" + jsonEscape(mats[m].file) + "\",\"name\":\"" + jsonEscape(mats[m].name) + "\",\"params\":[";
        bool first = true;
        for (const P& p : mats[m].params)
        {
            std::string type, value;
            char buf[64];
            auto floats = [&](size_t n) {
                std::string v = n > 1 ? "[" : "";
                for (size_t k = 0; k < n && (k + 1) * 4 <= p.value.size(); ++k)
                {
                    float x;
                    memcpy(&x, p.value.data() + 4 * k, 4);
                    snprintf(buf, sizeof(buf), "%s%.9g", k ? "," : "", (double)x);
                    v += buf;
                }
                return v + (n > 1 ? "]" : "");
            };
            switch (p.type)
            {
            case P::Type::eFloat: type = "float", value = floats(1); break;
            case P::Type::eFloat2: type = "float2", value = floats(2); break;
            case P::Type::eFloat3: type = "float3", value = floats(3); break;
            case P::Type::eFloat4: type = "float4", value = floats(4); break;
            case P::Type::eInt: {
                int v = 0;
                if (p.value.size() >= 4)
                    memcpy(&v, p.value.data(), 4);
                type = "int", value = std::to_string(v);
                break;
            }
            case P::Type::eBool: type = "bool", value = (!p.value.empty() && p.value[0]) ? "true" : "false"; break;
            case P::Type::eTexture:
                type = "texture", value = "\"" + jsonEscape(std::string(reinterpret_cast<const char*>(p.value.data()), p.value.size())) + "\"";
                break;
            }
            if (value.empty())
                continue;
            j += (first ? "{" : ",{");
            j += "\"name\":\"" + jsonEscape(p.name) + "\",\"type\":\"" + type + "\",\"value\":" + value + "}";
            first = false;
        }
        j += "]}";
    }
    return j + "]";
}
#endif
} // namespace skhdump

inline bool skhDumpScene(Scene& sc, const std::string& path)
{
    using namespace skhdump;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f)
        return false;
    const uint32_t nCameras = (uint32_t)sc.getCameraCount();
#ifdef SKH_WITH_STRELKA_HEADERS
    const uint32_t nTextures = 0; // (textures stay files: MDSC carries their paths, the loader resolves them beside the dump)
#else
    const uint32_t nTextures = (uint32_t)sc.getTextures().size();
#endif
    const uint32_t version = 1, sections = 10 + (nCameras ? 1u : 0u) + (nTextures ? 2u : 0u);
    fwrite("SKSCENE\0", 1, 8, f);
    fwrite(&version, 4, 1, f);
    fwrite(&sections, 4, 1, f);
    static_assert(sizeof(Scene::Vertex) == 32 && sizeof(Mesh) == 16 && sizeof(Curve) == 24 && sizeof(Scene::Light) == 112, "layouts (scene.h:21-42,80-89,146-155)");
    putSection(f, "VERT", 32, sc.getVertices().size(), sc.getVertices().data());
    putSection(f, "INDX", 4, sc.getIndices().size(), sc.getIndices().data());
    putSection(f, "MESH", 16, sc.getMeshes().size(), sc.getMeshes().data());
    putSection(f, "CPTS", 12, sc.getCurvesPoint().size(), sc.getCurvesPoint().data());
    putSection(f, "CWID", 4, sc.getCurvesWidths().size(), sc.getCurvesWidths().data());
    putSection(f, "CVCN", 4, sc.getCurvesVertexCounts().size(), sc.getCurvesVertexCounts().data());
    putSection(f, "CURV", 24, sc.getCurves().size(), sc.getCurves().data());
    std::vector<skh_instance> inst(sc.getInstances().size());
    for (size_t i = 0; i < inst.size(); ++i)
    {
        const Instance& in = sc.getInstances()[i];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c)
                inst[i].transform[4 * r + c] = in.transform[c][r]; // glm::float3x4(glm::rowMajor4(transform)), OptixRender.cpp:438
        inst[i].type = (uint32_t)in.type;
        inst[i].geom_id = in.mMeshId;
        inst[i].material_id = in.mMaterialId;
        inst[i].light_id = in.mLightId;
    }
    putSection(f, "INST", sizeof(skh_instance), inst.size(), inst.data());
    putSection(f, "LGHT", 112, sc.getLights().size(), sc.getLights().data());
#ifdef SKH_WITH_STRELKA_HEADERS
    const std::string mdsc = materialsToJson(sc.getMaterials());
    putSection(f, "MDSC", 1, mdsc.size(), mdsc.data());
#else
    std::vector<skh_material> mats;
    for (const Scene::MaterialDescription& m : sc.getMaterials())
        mats.push_back(m.args);
    putSection(f, "MATL", sizeof(skh_material), mats.size(), mats.data());
    if (nTextures)
    {
        std::vector<uint32_t> desc, texels;
        for (const Scene::Texture& t : sc.getTextures())
        {
            desc.insert(desc.end(), { (uint32_t)texels.size(), t.width, t.height, 0u });
            const size_t n = (size_t)t.width * t.height;
            texels.resize(texels.size() + n);
            memcpy(texels.data() + texels.size() - n, t.rgba8.data(), n * 4);
        }
        putSection(f, "TXDS", 16, nTextures, desc.data());
        putSection(f, "TXEL", 4, texels.size(), texels.data());
    }
#endif
    if (nCameras)
    {
        std::vector<CameraRecord> cams(nCameras);
        for (uint32_t k = 0; k < nCameras; ++k)
        {
            Camera& cam = sc.getCamera(k);
            memset(&cams[k], 0, sizeof(CameraRecord));
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c)
                    cams[k].view[4 * r + c] = cam.matrices.view[c][r];
            cams[k].fov = cam.fov;
            cams[k].znear = cam.znear;
            cams[k].zfar = cam.zfar;
        }
        putSection(f, "CAMR", sizeof(CameraRecord), cams.size(), cams.data());
    }
    const bool ok = !ferror(f);
    return fclose(f) == 0 && ok;
}

} // namespace oka
