// oka::HipBuffer / oka::HipRender: Strelka's render interface implemented on the C ABI of libstrelka_hip.so -- see HipRender.h.
// Plain C++17 host code (no HIP headers): every device operation goes through include/strelka_hip.h.  render() follows
// OptiXRender::render (src/render/optix/OptixRender.cpp:874-1057) statement by statement: the reset rules, the exposure formula, the
// sub-frame bookkeeping and the hand-back after the last sample are the adapter's contract.
#include "HipRender.h"
#ifdef SKH_WITH_STRELKA_HEADERS
#    include "SkhMaterials.h"

// The reference's texture loader (OptixRender.cpp:18,1191-1264).  In the reference STB_IMAGE_STATIC + STB_IMAGE_IMPLEMENTATION sit in OptixRender.cpp:16-17, which
// strelka_hip.cmake drops from the render target: this file fills that slot, so that targets which link `render` without the scene loader
// (HdStrelka: render + scene + materialmanager) still resolve stbi_load / stbi_image_free.
#    define STB_IMAGE_STATIC // (as OptixRender.cpp:16: the symbols stay file-local, gltfloader.cpp:5 has its own copy)
#    define STB_IMAGE_IMPLEMENTATION
#    include <stb_image.h>
#    include <filesystem>
#endif

#include <algorithm>
#include <cassert>
#include <cstdlib>
#include <cstring>
#include <iostream>

namespace oka
{

HipBuffer::HipBuffer(skh_context* ctx, void* devicePtr, BufferFormat format, uint32_t width, uint32_t height) : mCtx(ctx), mDeviceData(devicePtr)
{
    mFormat = format;
    mWidth = width;
    mHeight = height;
}
HipBuffer::~HipBuffer()
{
    if (mRegistered)
        skh_host_unregister(mCtx, mRegistered);
    if (mDeviceData)
        skh_buffer_free(mCtx, mDeviceData);
}
void HipBuffer::resize(uint32_t width, uint32_t height)
{
    if (mDeviceData)
        skh_buffer_free(mCtx, mDeviceData);
    mDeviceData = nullptr;
    mWidth = width;
    mHeight = height;
    skh_buffer_alloc(mCtx, (size_t)mWidth * mHeight * getElementSize(), &mDeviceData);
}
void* HipBuffer::map()
{
    const size_t bytes = (size_t)mWidth * mHeight * getElementSize();
    if (mHostData.size() != bytes || mRegistered != mHostData.data())
    {
        // the host mirror is the reference's std::vector (buffer.h:60-88); page-locked once per size so that the copy the caller
        // asks for after EVERY sub-frame (RenderPass.cpp:441-447) runs at PCIe rate instead of through a staging buffer
        if (mRegistered)
            skh_host_unregister(mCtx, mRegistered);
        mRegistered = nullptr;
        mHostData.resize(bytes);
        if (bytes && skh_host_register(mCtx, mHostData.data(), bytes) == SKH_OK)
            mRegistered = mHostData.data();
    }
    skh_buffer_download(mCtx, mDeviceData, mHostData.data(), bytes);
    return nullptr;
}

HipRender::~HipRender()
{
    if (mCtx && mGatherBuf)
        skh_buffer_free(mCtx, mGatherBuf);
    if (mCtx)
        skh_destroy(mCtx);
}

bool HipRender::enableTileSharing(const void* commId, int worldSize, int rank, uint32_t tileSize)
{
    if (!mCtx || worldSize < 1 || rank < 0 || rank >= worldSize)
        return false;
    if (worldSize > 1 && !check(skh_comm_init(mCtx, commId, worldSize, rank), "skh_comm_init"))
        return false;
    mSharing = true; // (world size 1 keeps the whole path -- tile set, gather, scatter -- minus the sends: what a 1-GPU box can test)
    mWorld = worldSize;
    mRank = rank;
    mTileSize = tileSize;
    mWidth = mHeight = 0; // the next render() re-derives the tile share
    return true;
}

// tiles t = rank (mod world) of the row-major tile list: interleaving spreads expensive image regions over the GPUs
bool HipRender::applyTiles(uint32_t width, uint32_t height)
{
    if (!mSharing)
        return true;
    std::vector<std::vector<uint32_t>> perRank(mWorld);
    uint32_t t = 0;
    for (uint32_t y = 0; y < height; y += mTileSize)
        for (uint32_t x = 0; x < width; x += mTileSize, ++t)
        {
            perRank[t % mWorld].push_back(x);
            perRank[t % mWorld].push_back(y);
        }
    mMaxTiles = (t + mWorld - 1) / mWorld;
    const std::vector<uint32_t>& mine = perRank[mRank];
    if (!check(skh_set_tiles(mCtx, mTileSize, mine.data(), (uint32_t)(mine.size() / 2)), "skh_set_tiles"))
        return false;
    if (mRank == 0)
    {
        // padding tiles get an origin outside the image: skh_scatter_tiles drops them
        mAllTileXY.assign((size_t)mWorld * mMaxTiles * 2, std::max(width, height));
        for (int r = 0; r < mWorld; ++r)
            std::copy(perRank[r].begin(), perRank[r].end(), mAllTileXY.begin() + (size_t)r * mMaxTiles * 2);
        if (mGatherBuf)
            skh_buffer_free(mCtx, mGatherBuf);
        mGatherBuf = nullptr;
        return check(skh_buffer_alloc(mCtx, (size_t)mWorld * mMaxTiles * mTileSize * mTileSize * 16, &mGatherBuf), "skh_buffer_alloc");
    }
    return true;
}
bool HipRender::check(skh_status s, const char* what)
{
    if (s == SKH_OK)
        return true;
    // reference behaviour: log + assert(0), keep going in release builds (OptixRender.cpp:61-103)
    mError = std::string(what) + ": " + (mCtx ? skh_last_error(mCtx) : "no context");
    std::cerr << "[HipRender] " << mError << std::endl;
    assert(0);
    return false;
}
void HipRender::init()
{
    int device = 0;
    if (const char* lr = getenv("LOCAL_RANK"))
        device = atoi(lr);
    check(skh_create(device, &mCtx), "skh_create");
}
Buffer* HipRender::createBuffer(const BufferDesc& desc)
{
    assert(desc.format == BufferFormat::FLOAT4); // OptixRender.cpp:1109
    void* d = nullptr;
    if (!check(skh_buffer_alloc(mCtx, (size_t)desc.width * desc.height * Buffer::getElementSize(desc.format), &d), "skh_buffer_alloc"))
        return nullptr;
    return new HipBuffer(mCtx, d, desc.format, desc.width, desc.height);
}
void HipRender::uploadScene()
{
    Scene& sc = *mScene;
    static_assert(sizeof(Scene::Vertex) == sizeof(skh_vertex) && sizeof(Mesh) == sizeof(skh_mesh) && sizeof(Curve) == sizeof(skh_curve), "layouts");
    static_assert(sizeof(Scene::Light) == sizeof(skh_light), "light layout");
    check(skh_set_geometry(mCtx, reinterpret_cast<const skh_vertex*>(sc.getVertices().data()), (uint32_t)sc.getVertices().size(),
                           sc.getIndices().data(), (uint32_t)sc.getIndices().size(),
                           reinterpret_cast<const skh_mesh*>(sc.getMeshes().data()), (uint32_t)sc.getMeshes().size()),
          "skh_set_geometry");
    if (!sc.getCurves().empty())
        check(skh_set_curves(mCtx, reinterpret_cast<const float*>(sc.getCurvesPoint().data()), (uint32_t)sc.getCurvesPoint().size(),
                             sc.getCurvesWidths().data(), (uint32_t)sc.getCurvesWidths().size(), sc.getCurvesVertexCounts().data(),
                             (uint32_t)sc.getCurvesVertexCounts().size(), reinterpret_cast<const skh_curve*>(sc.getCurves().data()),
                             (uint32_t)sc.getCurves().size()),
              "skh_set_curves");
    std::vector<skh_instance> inst(sc.getInstances().size());
    for (size_t i = 0; i < inst.size(); ++i)
    {
        const Instance& in = sc.getInstances()[i];
        // glm::float3x4(glm::rowMajor4(transform)) (OptixRender.cpp:438): rows of the affine transform
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c)
                inst[i].transform[4 * r + c] = in.transform[c][r];
        inst[i].type = (uint32_t)in.type;
        inst[i].geom_id = in.mMeshId;
        inst[i].material_id = in.mMaterialId;
        inst[i].light_id = in.mLightId;
    }
    check(skh_set_instances(mCtx, inst.data(), (uint32_t)inst.size()), "skh_set_instances");
    check(skh_set_lights(mCtx, reinterpret_cast<const skh_light*>(sc.getLights().data()), (uint32_t)sc.getLights().size()), "skh_set_lights");
    uploadMaterials();
    check(skh_build_accel(mCtx, SKH_BUILD_LBVH), "skh_build_accel");
}

void HipRender::uploadMaterials()
{
    Scene& sc = *mScene;
    std::vector<skh_material> mats;
#ifdef SKH_WITH_STRELKA_HEADERS
    // every eTexture parameter becomes one RGBA8 texture (OptixRender.cpp:1346-1377: resolved against resource/searchPath, stbi_load
    // with STBI_rgb_alpha); a file that cannot be read is reported and the material keeps its constant colour (:1195-1199)
    std::vector<std::vector<uint8_t>> pixels;
    std::vector<skh_texture> tex;
    const std::filesystem::path searchPath = getSharedContext().mSettingsManager->getAs<std::string>("resource/searchPath");
    auto load = [&](const std::string& rel) -> uint32_t {
        if (rel.empty())
            return 0u;
        int w = 0, h = 0, ch = 0;
        stbi_uc* data = stbi_load((searchPath / rel).string().c_str(), &w, &h, &ch, STBI_rgb_alpha);
        if (!data)
        {
            std::cerr << "[HipRender] unable to load texture from file: " << rel << std::endl;
            return 0u;
        }
        pixels.emplace_back(data, data + (size_t)w * h * 4);
        stbi_image_free(data);
        tex.push_back(skh_texture{ nullptr, (uint32_t)w, (uint32_t)h });
        return (uint32_t)tex.size(); // 1-based, 0 = none (MDL's resource numbering: texture_support_cuda.h:300-304)
    };
    for (const Scene::MaterialDescription& d : sc.getMaterials())
    {
        const uint32_t diffuseId = load(skhmat::texturePath(d, "diffuse_texture")), normalId = load(skhmat::texturePath(d, "normalmap_texture"));
        mats.push_back(skhmat::translate(d, diffuseId, normalId));
    }
    for (size_t k = 0; k < tex.size(); ++k)
        tex[k].rgba8 = pixels[k].data();
#else
    std::vector<skh_texture> tex; // OptixRender.cpp:1352-1377: one texture object per eTexture parameter
    for (const Scene::Texture& t : sc.getTextures())
        tex.push_back(skh_texture{ t.rgba8.data(), t.width, t.height });
    for (const auto& m : sc.getMaterials())
        mats.push_back(m.args);
#endif
    check(skh_set_textures(mCtx, tex.data(), (uint32_t)tex.size()), "skh_set_textures");
    if (mats.empty())
    {
        skh_material m; // default.mdl::default_material registered by init() in the reference (OptixRender.cpp:1090-1097)
        memset(&m, 0, sizeof(m));
        m.base_color[0] = m.base_color[1] = m.base_color[2] = 0.8f;
        mats.push_back(m);
    }
    check(skh_set_materials(mCtx, mats.data(), (uint32_t)mats.size()), "skh_set_materials");
}

void HipRender::render(Buffer* output)
{
    SharedContext& sh = getSharedContext();
    if (sh.mFrameNumber == 0)
        uploadScene(); // scene is uploaded once; later edits are ignored, like the reference (OptixRender.cpp:876-888)

    const uint32_t width = output->width();
    const uint32_t height = output->height();
    // updatePathtracerParams (OptixRender.cpp:827-872)
    if (mWidth != width || mHeight != height)
    {
        sh.mSubframeIndex = 0;
        sh.mSettingsManager->setAs<bool>("render/pt/isResized", true);
        applyTiles(width, height);
        check(skh_resize(mCtx, width, height), "skh_resize");
        mWidth = width;
        mHeight = height;
    }
    Camera& camera = mScene->getCamera(0);
    camera.updateAspectRatio(width / (float)height);
    camera.updateViewMatrix();
    if (camera.matrices.perspective != mPrevPerspective || camera.matrices.view != mPrevView)
        sh.mSubframeIndex = 0; // need reset (OptixRender.cpp:903-908)

    SettingsManager& settings = *sh.mSettingsManager;
    bool settingsChanged = false;
    const uint32_t rectLightSamplingMethod = settings.getAs<uint32_t>("render/pt/rectLightSamplingMethod");
    settingsChanged = (mRectLightSamplingMethodPrev != rectLightSamplingMethod);
    mRectLightSamplingMethodPrev = rectLightSamplingMethod;
    bool enableAccumulation = settings.getAs<bool>("render/pt/enableAcc");
    settingsChanged |= (mEnableAccumulationPrev != enableAccumulation);
    mEnableAccumulationPrev = enableAccumulation;
    const uint32_t sspTotal = settings.getAs<uint32_t>("render/pt/sppTotal");
    settingsChanged |= (mSppTotalPrev > sspTotal); // reset only if the new spp is below what was already accumulated
    mSppTotalPrev = sspTotal;
    const float gamma = settings.getAs<float>("render/post/gamma");
    const uint32_t tonemapperType = settings.getAs<uint32_t>("render/pt/tonemapperType");
    if (settingsChanged)
        sh.mSubframeIndex = 0;

    skh_frame_params p;
    memset(&p, 0, sizeof(p));
    p.max_depth = settings.getAs<uint32_t>("render/pt/depth");
    p.rect_light_sampling_method = rectLightSamplingMethod;
    p.debug = settings.getAs<uint32_t>("render/pt/debug");
    p.shadow_ray_tmin = settings.getAs<float>("render/pt/dev/shadowRayTmin");
    p.material_ray_tmin = settings.getAs<float>("render/pt/dev/materialRayTmin");
    // glm::transpose(glm::inverse(view)) / glm::transpose(invPerspective) memcpy'd column-major == row-major matrices
    const Mat4 v2w = inverse(camera.matrices.view); // (glm::inverse / oka::inverse, found by ADL)
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
        {
            p.view_to_world[4 * r + c] = v2w[c][r];
            p.clip_to_view[4 * r + c] = camera.matrices.invPerspective[c][r];
        }
    p.subframe_index = (uint32_t)sh.mSubframeIndex;
    // photometric exposure (OptixRender.cpp:961-987)
    const float filmIso = settings.getAs<float>("render/post/tonemapper/filmIso");
    const float cm2_factor = settings.getAs<float>("render/post/tonemapper/cm2_factor");
    const float fStop = settings.getAs<float>("render/post/tonemapper/fStop");
    const float shutterSpeed = settings.getAs<float>("render/post/tonemapper/shutterSpeed");
    float e[3] = { 1.0f, 1.0f, 1.0f };
    const float lum = e[0] * 0.299f + e[1] * 0.587f + e[2] * 0.114f;
    const float k = filmIso > 0.0f ? cm2_factor * filmIso / (shutterSpeed * fStop * fStop) / 100.0f : cm2_factor;
    const float invLum = 1.0f / lum;
    for (int i = 0; i < 3; ++i)
        p.exposure[i] = e[i] * k * invLum;

    const uint32_t totalSpp = sspTotal;
    const uint32_t samplesPerLaunch = settings.getAs<uint32_t>("render/pt/spp");
    const int32_t leftSpp = (int32_t)totalSpp - (int32_t)sh.mSubframeIndex;
    uint32_t samplesThisLaunch = enableAccumulation ? (uint32_t)std::min((int32_t)samplesPerLaunch, leftSpp) : samplesPerLaunch;
    if (p.debug == 1)
    {
        samplesThisLaunch = 1;
        enableAccumulation = false;
    }
    p.samples_this_launch = samplesThisLaunch;
    p.enable_accumulation = enableAccumulation;
    p.spp_total = totalSpp;

    void* dImage = static_cast<HipBuffer*>(output)->getNativePtr();
    if (samplesThisLaunch != 0)
    {
        check(skh_render_subframe(mCtx, &p, dImage), "skh_render_subframe");
        if (enableAccumulation)
            sh.mSubframeIndex += samplesThisLaunch;
        else
            sh.mSubframeIndex = 0;
    }
    else
    {
        // all spp done: the latest accumulated RAW radiance -- or, for the debug views 2 / 3, the raw diffuse / specular AOV -- goes back
        // to the image before every tonemap() below (OptixRender.cpp:1022-1043); without the copy the in-place tonemap would be applied to
        // an already tonemapped image call after call
        if (p.debug == 0)
            check(skh_copy_accum(mCtx, dImage), "skh_copy_accum");
        else if (p.debug == 2 || p.debug == 3)
            check(skh_copy_aov(mCtx, p.debug - 2, dImage), "skh_copy_aov");
    }
    if (mSharing && p.debug == 0 && enableAccumulation)
    {
        // the frame's one collective: every rank's tile accumulators -> rank 0, whose output then holds the whole image
        check(skh_gather_tiles(mCtx, mMaxTiles, mRank == 0 ? mGatherBuf : nullptr, 0), "skh_gather_tiles");
        if (mRank == 0)
            check(skh_scatter_tiles(mCtx, mGatherBuf, mAllTileXY.data(), (uint32_t)(mAllTileXY.size() / 2), mTileSize, dImage, width, height),
                  "skh_scatter_tiles");
    }
    if (p.debug != 1)
        check(skh_tonemap(mCtx, dImage, width, height, tonemapperType, p.exposure, gamma), "skh_tonemap");
    output->unmap();
    sh.mFrameNumber++;
    mPrevView = camera.matrices.view;
    mPrevPerspective = camera.matrices.perspective;
}

#ifndef SKH_WITH_STRELKA_HEADERS
// (in the Strelka tree RenderFactory lives in src/render/render.cpp: integration/strelka_hip.patch adds the eCompute branch there)
Render* RenderFactory::createRender(RenderType type)
{
    if (type == RenderType::eCompute)
        return new HipRender();
    return nullptr; // eOptiX / eMetal live in the reference tree
}
Render* RenderFactory::createRender()
{
    return new HipRender();
}
#endif

} // namespace oka
