// oka::HipRender -- the MI355X backend behind Strelka's render interface, in the RenderType::eCompute slot the reference declares but
// never implements (include/render/render.h:9-14, src/render/render.cpp:10-26).  What OptiXRender / OptixBuffer are to OptiX
// (src/render/optix/OptixRender.{h,cpp}, OptixBuffer.{h,cpp}), these two classes are to libstrelka_hip.so (include/strelka_hip.h).
//
// ONE source, two sets of headers (a host-side include switch only -- the device code has a single path):
//   -DSKH_WITH_STRELKA_HEADERS   inside the Strelka tree: <render/render.h>, <render/buffer.h>, <scene/scene.h>, <settings/settings.h>
//                                 (glm types; integration/strelka_hip.cmake + strelka_hip.patch wire it in)
//   otherwise                     strelka_amd/host/oka_mirror.h, this repository's dependency-free stand-in for those headers
//                                 (what the tests and the benchmark box compile: glm / MDL SDK / OpenUSD are not in this image)
// The code below uses only what both provide: m[col][row] element access, inverse(m) found by ADL, operator!= on matrices, the
// scene's flat-array getters, SettingsManager::getAs / setAs.
#pragma once
#ifdef SKH_WITH_STRELKA_HEADERS
#    include <render/render.h>
#    include <render/buffer.h>
#    include <scene/scene.h>
#    include <settings/settings.h>
#    include <strelka_hip.h>
#else
#    include "../strelka_amd/host/oka_mirror.h"
#endif

#include <string>
#include <vector>

namespace oka
{

class HipBuffer : public Buffer // OptixBuffer.{h,cpp}
{
public:
    HipBuffer(skh_context* ctx, void* devicePtr, BufferFormat format, uint32_t width, uint32_t height);
    ~HipBuffer() override;
    void resize(uint32_t width, uint32_t height) override;
    void* map() override; // D2H into mHostData, returns nullptr like OptixBuffer::map (OptixBuffer.cpp:37-43)
    void unmap() override
    {
    }
    void* getNativePtr()
    {
        return mDeviceData;
    }

private:
    skh_context* mCtx;
    void* mDeviceData = nullptr;
    void* mRegistered = nullptr; // mHostData's storage while it is page-locked (skh_host_register)
};

class HipRender : public Render
{
public:
    using Mat4 = decltype(Camera::Matrices::view); // glm::float4x4 in the Strelka tree, oka::float4x4 in the mirror

    HipRender() = default;
    ~HipRender() override;
    void init() override; // OptiXRender::init (OptixRender.cpp:1059-1105): context + default material 0
    void render(Buffer* output) override; // OptiXRender::render (OptixRender.cpp:874-1057)
    Buffer* createBuffer(const BufferDesc& desc) override; // OptixRender.cpp:1107-1115
    void* getNativeDevicePtr() override
    {
        return mCtx;
    }
    const std::string& lastError() const
    {
        return mError;
    }
    skh_context* context()
    {
        return mCtx;
    }
    // Multi-GPU (new: the reference is one process on one GPU).  One HipRender per process per GPU; this one renders the pixel
    // tiles t = rank (mod worldSize) of every frame -- the split is by tile because the accumulator is an order-dependent LDR-space
    // lerp per pixel (OptixRender.cu:60-78): sharding by samples would change the image -- and after every render() the tile
    // accumulators of all ranks are gathered to rank 0 below the C ABI (skh_gather_tiles: RCCL sends over xGMI), whose output
    // buffer then holds the whole frame (the other ranks' buffers hold their own tiles).  commId: 128 bytes from
    // skh_comm_unique_id on rank 0, handed to every rank by whatever launched the processes.  Call after init(), before render().
    bool enableTileSharing(const void* commId, int worldSize, int rank, uint32_t tileSize = 32);

private:
    skh_context* mCtx = nullptr;
    std::string mError;
    uint32_t mWidth = 0, mHeight = 0;
    Mat4 mPrevView{ 0.0f }, mPrevPerspective{ 0.0f };
    // per-instance "previous settings" (function-static in the reference, which makes it non re-entrant:
    // OptixRender.cpp:913,918,923)
    uint32_t mRectLightSamplingMethodPrev = 0, mSppTotalPrev = 0;
    bool mEnableAccumulationPrev = false;
    bool check(skh_status s, const char* what);
    bool applyTiles(uint32_t width, uint32_t height); // this rank's share of the frame -> skh_set_tiles (multi-GPU)
    bool mSharing = false;
    int mWorld = 1, mRank = 0;
    uint32_t mTileSize = 32, mMaxTiles = 0;
    std::vector<uint32_t> mAllTileXY; // root: (x0, y0) of every rank's tiles, rank-major, padded to mMaxTiles per rank
    void* mGatherBuf = nullptr; // root: [world][mMaxTiles][tile^2] float4
    void uploadScene(); // mFrameNumber == 0 block: OptixRender.cpp:876-888
    void uploadMaterials(); // MaterialDescription list -> skh_material blocks + textures (OptixRender.cpp:1270-1433)
};

} // namespace oka
