// TEST SCAFFOLDING: runs integration/SkSceneDump.h's REAL-header branch (materials written as the MDSC JSON section) on the look-alike headers of
// tests/cpp/strelka_lookalike/ and leaves the file for tests/test_integration_files.py to read back with strelka_amd/scene_io.py.
#include "../../integration/SkSceneDump.h"

#include <cstring>

static oka::MaterialManager::Param param(const char* name, oka::MaterialManager::Param::Type t, const void* p, size_t n)
{
    oka::MaterialManager::Param q;
    q.name = name;
    q.type = t;
    q.value.assign((const uint8_t*)p, (const uint8_t*)p + n);
    return q;
}

int main(int argc, char** argv)
{
    using P = oka::MaterialManager::Param;
    oka::Scene sc;
    {
        oka::Scene::MaterialDescription d;
        d.file = "default.mdl", d.name = "default_material";
        const float c[3] = { 0.25f, 0.5f, 0.75f };
        d.params.push_back(param("diffuse_color", P::Type::eFloat3, c, sizeof(c)));
        sc.getMaterials().push_back(d);
    }
    {
        oka::Scene::MaterialDescription d;
        d.file = "OmniPBR.mdl", d.name = "OmniPBR";
        const float c[3] = { 0.9f, 0.1f, 0.2f }, r = 0.35f, m = 1.0f;
        const char* tex = "textures/wood \"oak\".png";
        d.params.push_back(param("diffuse_color_constant", P::Type::eFloat3, c, sizeof(c)));
        d.params.push_back(param("reflection_roughness_constant", P::Type::eFloat, &r, 4));
        d.params.push_back(param("metallic_constant", P::Type::eFloat, &m, 4));
        d.params.push_back(param("diffuse_texture", P::Type::eTexture, tex, strlen(tex)));
        sc.getMaterials().push_back(d);
    }
    {
        oka::Scene::MaterialDescription d;
        d.file = "OmniGlass.mdl", d.name = "OmniGlass";
        const float ior = 1.33f, fr = 0.4f, c4[4] = { 1, 2, 3, 4 }, c2[2] = { 5, 6 };
        const uint8_t yes = 1;
        const int depth = 7;
        d.params.push_back(param("glass_ior", P::Type::eFloat, &ior, 4));
        d.params.push_back(param("frosting_roughness", P::Type::eFloat, &fr, 4));
        d.params.push_back(param("thin_walled", P::Type::eBool, &yes, 1));
        d.params.push_back(param("depth", P::Type::eInt, &depth, 4));
        d.params.push_back(param("some_float4", P::Type::eFloat4, c4, sizeof(c4)));
        d.params.push_back(param("some_float2", P::Type::eFloat2, c2, sizeof(c2)));
        sc.getMaterials().push_back(d);
    }
    return argc > 1 && oka::skhDumpScene(sc, argv[1]) ? 0 : 1;
}
