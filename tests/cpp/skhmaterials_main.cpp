// Runs integration/SkhMaterials.h's translation on a LOCAL look-alike of the reference's two structs (oka::Scene::MaterialDescription,
// scene.h:65-78; oka::MaterialManager::Param, materialmanager.h:33-48 -- field names, types and the Type numbering as there), so that the
// C++ statement can be compared case by case with strelka_amd/scene_io.py::material_from_description without the reference headers
// (which need glm).  Test scaffolding: nothing in the product includes this file.
// stdin: cases as text -- "D <file>|<name>|<n params>" then per parameter "P <type> <name> <hex bytes>"; stdout: one 64-byte skh_material per case.
#include "../../integration/SkhMaterials.h"

#include <cstdio>
#include <iostream>
#include <sstream>
#include <vector>

struct Param
{
    enum class Type : uint32_t
    {
        eFloat = 0,
        eInt,
        eBool,
        eFloat2,
        eFloat3,
        eFloat4,
        eTexture
    };
    Type type;
    std::string name;
    std::vector<uint8_t> value;
};
struct MaterialDescription
{
    std::string code, file, name;
    bool hasColor = false;
    float color[3] = { 0, 0, 0 };
    std::vector<Param> params;
};

int main()
{
    std::string line;
    std::vector<MaterialDescription> all;
    while (std::getline(std::cin, line))
    {
        if (line.rfind("D ", 0) == 0)
        {
            MaterialDescription d;
            const size_t a = line.find('|'), b = line.find('|', a + 1);
            d.file = line.substr(2, a - 2);
            d.name = line.substr(a + 1, b - a - 1);
            all.push_back(d);
        }
        else if (line.rfind("P ", 0) == 0)
        {
            std::istringstream is(line.substr(2));
            uint32_t type;
            std::string name, hex;
            is >> type >> name >> hex;
            Param p;
            p.type = (Param::Type)type;
            p.name = name;
            if (hex != "-")
                for (size_t k = 0; k + 1 < hex.size(); k += 2)
                    p.value.push_back((uint8_t)std::stoul(hex.substr(k, 2), nullptr, 16));
            all.back().params.push_back(p);
        }
    }
    for (const MaterialDescription& d : all)
    {
        const bool dt = !oka::skhmat::texturePath(d, "diffuse_texture").empty(), nt = !oka::skhmat::texturePath(d, "normalmap_texture").empty();
        const skh_material m = oka::skhmat::translate(d, dt ? 1u : 0u, nt ? (dt ? 2u : 1u) : 0u);
        fwrite(&m, sizeof(m), 1, stdout);
    }
    return 0;
}
