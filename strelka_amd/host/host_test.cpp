// Exercises the C++ host mirror (oka_render.h) the way hdRunner's frame loop drives the reference
// (src/hdRunner/main.cpp:500-763): settings defaults -> RenderFactory::createRender(eCompute) -> init -> loop
// { render(outputBuffer); map() }.
//   host_test cpu <outdir>   builds the test scene through the oka::Scene API and dumps the flat arrays (no GPU needed)
//   host_test gpu <outdir> <frames>   additionally renders <frames> frames and dumps the mapped image
//   host_test gpu-aov2 | gpu-aov3 <outdir> <frames>   the same with the diffuse / specular AOV debug view, plus five render() calls after the last sample
#include "oka_render.h"
#include "../../integration/SkSceneDump.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

using namespace oka;

template <typename T>
static void dump(const std::string& dir, const char* name, const T* data, size_t n)
{
    FILE* f = fopen((dir + "/" + name).c_str(), "wb");
    if (!f)
    {
        perror(name);
        exit(2);
    }
    fwrite(data, sizeof(T), n, f);
    fclose(f);
}

static std::vector<Scene::Vertex> quad(const float3 p[4], const float3& n)
{
    // de-indexed like HdStrelkaMesh::_UpdateGeometry: 3 fresh vertices per triangle (Mesh.cpp:140-178)
    const int idx[6] = { 0, 1, 2, 0, 2, 3 };
    std::vector<Scene::Vertex> vb(6);
    for (int i = 0; i < 6; ++i)
    {
        vb[i] = Scene::Vertex{};
        vb[i].pos = p[idx[i]];
        vb[i].normal = packNormals(n);
        vb[i].tangent = packNormals(float3{ 1, 0, 0 });
        vb[i].uv = packUV(0.0f, 0.0f);
    }
    return vb;
}

static void buildScene(Scene& sc)
{
    Scene::MaterialDescription white{ "default_material", {} };
    memset(&white.args, 0, sizeof(white.args));
    white.args.type = SKH_MAT_DIFFUSE;
    white.args.base_color[0] = white.args.base_color[1] = white.args.base_color[2] = 0.8f;
    Scene::MaterialDescription pbr = white;
    pbr.name = "OmniPBR";
    pbr.args.type = SKH_MAT_PBR;
    pbr.args.base_color[0] = 0.9f, pbr.args.base_color[1] = 0.3f, pbr.args.base_color[2] = 0.2f;
    pbr.args.roughness = 0.3f, pbr.args.metallic = 0.0f, pbr.args.specular = 0.5f, pbr.args.ior = 1.5f;
    const uint32_t m0 = sc.addMaterial(white), m1 = sc.addMaterial(pbr);
    const float3 fl[4] = { { -2, 0, 2 }, { 2, 0, 2 }, { 2, 0, -2 }, { -2, 0, -2 } };
    const uint32_t floorMesh = sc.createMesh(quad(fl, float3{ 0, 1, 0 }), { 0, 1, 2, 3, 4, 5 });
    const float3 pl[4] = { { -0.5f, 0, 0 }, { 0.5f, 0, 0 }, { 0.5f, 1, 0 }, { -0.5f, 1, 0 } };
    const uint32_t panel = sc.createMesh(quad(pl, float3{ 0, 0, 1 }), { 0, 1, 2, 3, 4, 5 });
    sc.createInstance(Instance::Type::eMesh, floorMesh, m0, float4x4(1.0f));
    const float4x4 xf = float4x4::translate(float3{ 0.3f, 0.0f, -0.4f }) *
                        float4x4::fromQuat(quatFromEulerRadians(float3{ 0.0f, 0.6f, 0.0f })) * float4x4::scale(float3{ 1.5f, 1.2f, 1.0f });
    sc.createInstance(Instance::Type::eMesh, panel, m1, xf);
    Scene::UniformLightDesc rect;
    rect.type = 0;
    rect.useXform = true;
    rect.xform = float4x4::translate(float3{ 0.0f, 2.5f, 0.5f }) * float4x4::fromQuat(quatFromEulerRadians(float3{ -1.5707963f, 0, 0 }));
    rect.width = 0.8f, rect.height = 0.6f, rect.color = float3{ 1.0f, 0.9f, 0.8f }, rect.intensity = 40.0f;
    sc.createLight(rect);
    Scene::UniformLightDesc sph;
    sph.type = 2;
    sph.useXform = false;
    sph.position = float3{ -1.2f, 0.8f, 0.6f };
    sph.orientation = float3{ 10.0f, 20.0f, 30.0f };
    sph.radius = 0.15f, sph.color = float3{ 0.5f, 0.7f, 1.0f }, sph.intensity = 25.0f;
    sc.createLight(sph);
    Scene::UniformLightDesc dist;
    dist.type = 3;
    dist.useXform = true;
    dist.xform = float4x4::fromQuat(quatFromEulerRadians(float3{ -0.9f, 0.4f, 0.0f }));
    dist.halfAngle = 0.0872664626f, dist.color = float3{ 1, 1, 1 }, dist.intensity = 1.5f, dist.radius = 0.0f;
    sc.createLight(dist);
    Camera cam;
    cam.fov = 50.0f;
    cam.lookAt(float3{ 1.5f, 1.8f, 3.5f }, float3{ 0.0f, 0.6f, 0.0f }, float3{ 0, 1, 0 });
    sc.addCamera(cam);
}

int main(int argc, char** argv)
{
    if (argc < 3)
    {
        fprintf(stderr, "usage: host_test cpu|gpu <outdir> [frames] | host_test load <outdir> <file.skscene>\n");
        return 2;
    }
    const std::string mode = argv[1], dir = argv[2];
    int frames = argc > 3 ? atoi(argv[3]) : 6;
    if (mode == "load")
    {
        // host_test load <outdir> <file.skscene>: read a dump (e.g. one written by strelka_amd/scene_io.py) and write it back
        Scene loaded;
        if (argc < 4 || !loaded.loadDump(argv[3]))
        {
            fprintf(stderr, "loadDump failed\n");
            return 6;
        }
        if (!loaded.saveDump(dir + "/resaved.skscene"))
            return 7;
        printf("host_test load ok: %zu vertices, %zu instances, %zu lights, %zu cameras\n", loaded.getVertices().size(), loaded.getInstances().size(),
               loaded.getLights().size(), loaded.getCameraCount());
        return 0;
    }
    Scene scene;
    buildScene(scene);
    if (!scene.saveDump(dir + "/scene.skscene"))
        return 8;
    // the exporter a Strelka tree runs (integration/SkSceneDump.h, strelka_hip.patch) must write the same file
    if (!skhDumpScene(scene, dir + "/scene_exporter.skscene"))
        return 9;
    dump(dir, "vertices.bin", scene.getVertices().data(), scene.getVertices().size());
    dump(dir, "indices.bin", scene.getIndices().data(), scene.getIndices().size());
    dump(dir, "meshes.bin", scene.getMeshes().data(), scene.getMeshes().size());
    dump(dir, "lights.bin", scene.getLights().data(), scene.getLights().size());
    std::vector<skh_instance> inst(scene.getInstances().size());
    for (size_t i = 0; i < inst.size(); ++i)
    {
        const Instance& in = scene.getInstances()[i];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c)
                inst[i].transform[4 * r + c] = in.transform.m[c][r];
        inst[i].type = (uint32_t)in.type, inst[i].geom_id = in.mMeshId, inst[i].material_id = in.mMaterialId, inst[i].light_id = in.mLightId;
    }
    dump(dir, "instances.bin", inst.data(), inst.size());
    std::vector<skh_material> mats;
    for (auto& m : scene.getMaterials())
        mats.push_back(m.args);
    dump(dir, "materials.bin", mats.data(), mats.size());
    const uint32_t W = 96, H = 64;
    Camera& cam = scene.getCamera(0);
    cam.updateAspectRatio(W / (float)H);
    cam.updateViewMatrix();
    float mtx[32];
    const float4x4 v2w = cam.matrices.view.inverse();
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
        {
            mtx[4 * r + c] = v2w.m[c][r];
            mtx[16 + 4 * r + c] = cam.matrices.invPerspective.m[c][r];
        }
    dump(dir, "camera.bin", mtx, 32);
    // SettingsManager semantics
    SettingsManager sm;
    sm.setAs<uint32_t>("render/pt/depth", 4);
    sm.setAs<float>("render/post/gamma", 2.4f);
    sm.setAs<bool>("render/pt/enableAcc", true);
    if (sm.getAs<uint32_t>("render/pt/depth") != 4 || std::fabs(sm.getAs<float>("render/post/gamma") - 2.4f) > 1e-6f || !sm.getAs<bool>("render/pt/enableAcc"))
        return 3;
    if (RenderFactory::createRender(RenderType::eOptiX) != nullptr)
        return 4; // only the eCompute slot is served here
    if (mode == "cpu")
    {
        printf("host_test cpu ok: %zu vertices, %zu instances, %zu lights\n", scene.getVertices().size(), inst.size(), scene.getLights().size());
        return 0;
    }
    // ---- hdRunner defaults (src/hdRunner/main.cpp:510-542) ----
    SharedContext ctx;
    ctx.mSettingsManager = &sm;
    sm.setAs<uint32_t>("render/width", W);
    sm.setAs<uint32_t>("render/height", H);
    sm.setAs<uint32_t>("render/pt/depth", 4);
    sm.setAs<uint32_t>("render/pt/sppTotal", (uint32_t)frames - 1); // the last frame exercises the "all spp done" copy path
    sm.setAs<uint32_t>("render/pt/spp", 1);
    sm.setAs<uint32_t>("render/pt/tonemapperType", 1);
    // gpu-aov2 / gpu-aov3: the debug views 2 / 3 (diffuse / specular AOV as the image, OptixRender.cu:169-247)
    const uint32_t debugView = mode == "gpu-aov2" ? 2u : (mode == "gpu-aov3" ? 3u : 0u);
    sm.setAs<uint32_t>("render/pt/debug", debugView);
    sm.setAs<bool>("render/pt/enableAcc", true);
    sm.setAs<bool>("render/pt/isResized", false);
    sm.setAs<uint32_t>("render/pt/rectLightSamplingMethod", 0);
    sm.setAs<float>("render/post/tonemapper/filmIso", 100.0f);
    sm.setAs<float>("render/post/tonemapper/cm2_factor", 1.0f);
    sm.setAs<float>("render/post/tonemapper/fStop", 4.0f);
    sm.setAs<float>("render/post/tonemapper/shutterSpeed", 100.0f);
    sm.setAs<float>("render/post/gamma", 2.4f);
    sm.setAs<float>("render/pt/dev/shadowRayTmin", 0.0f);
    sm.setAs<float>("render/pt/dev/materialRayTmin", 0.0f);
    Render* render = RenderFactory::createRender(RenderType::eCompute);
    render->setSharedContext(&ctx);
    ctx.mRender = render;
    render->setScene(&scene);
    render->init();
    HipRender* hr = static_cast<HipRender*>(render);
    if (!hr->lastError().empty())
    {
        fprintf(stderr, "init failed: %s\n", hr->lastError().c_str());
        return 5;
    }
    if (mode == "gpu-tiles")
    {
        // the multi-GPU path with the communicator a 1-GPU box can form (world size 1): tile set, gather below the C ABI, scatter
        unsigned char id[SKH_COMM_ID_BYTES];
        if (skh_comm_unique_id(id) != SKH_OK || !hr->enableTileSharing(id, 1, 0, 16))
        {
            fprintf(stderr, "enableTileSharing failed: %s\n", hr->lastError().c_str());
            return 9;
        }
    }
    Buffer* out = render->createBuffer(BufferDesc{ W, H, BufferFormat::FLOAT4 });
    for (int f = 0; f < frames; ++f)
    {
        render->render(out);
        out->map();
    }
    if (debugView)
    {
        // all samples are done: every further render() must hand back the same picture -- the raw AOV copied to the image, then
        // tonemapped (OptixRender.cpp:1029-1049) -- not tonemap the already tonemapped image again
        dump(dir, "image_first.bin", (const char*)out->getHostPointer(), out->getHostDataSize());
        for (int f = 0; f < 5; ++f)
        {
            render->render(out);
            out->map();
        }
        std::vector<float> aov((size_t)W * H * 4);
        skh_read_aov(hr->context(), debugView - 2u, aov.data());
        dump(dir, "aov.bin", aov.data(), aov.size());
        frames += 5;
    }
    dump(dir, "image.bin", (const char*)out->getHostPointer(), out->getHostDataSize());
    std::vector<float> accum((size_t)W * H * 4);
    skh_read_accum(hr->context(), accum.data());
    dump(dir, "accum.bin", accum.data(), accum.size());
    printf("host_test gpu ok: frames %d, subframeIndex %zu, frameNumber %zu\n", frames, ctx.mSubframeIndex, ctx.mFrameNumber);
    if (ctx.mSubframeIndex != (size_t)sm.getAs<uint32_t>("render/pt/sppTotal") || ctx.mFrameNumber != (size_t)frames || !hr->lastError().empty())
        return 6;
    delete out;
    delete render;
    return 0;
}
