// STAND-IN for the reference's headers (oka_mirror.h): the types oka::HipRender (integration/HipRender.{h,cpp}) is written against,
// for builds WITHOUT the Strelka tree (this repo's tests, the benchmark box).  Inside the Strelka tree the same HipRender sources are
// compiled with -DSKH_WITH_STRELKA_HEADERS and include <render/render.h> / <scene/scene.h> instead of this file.
//
// Host-side mirror of the reference's render interface for the hot path, above the C ABI (include/strelka_hip.h).
//
// Same names, argument meaning and error behaviour as the reference headers, so that code written against Strelka's
// API (HdStrelka's render pass, the glTF app loop) reads the same against this backend:
//   oka::Render / RenderType / RenderFactory     include/render/render.h:9-63, src/render/render.cpp:10-35
//   oka::Buffer / BufferDesc / BufferFormat      include/render/buffer.h:9-88, src/render/optix/OptixBuffer.cpp
//   oka::SharedContext / Result                  include/render/common.h:22-35
//   oka::SettingsManager                         include/settings/settings.h:11-118
//   oka::Scene / Mesh / Curve / Instance / Light include/scene/scene.h, src/scene/scene.cpp
//   oka::Camera                                  include/scene/camera.h, src/scene/camera.cpp
// The reference's headers pull in glm / MDL SDK types; this mirror carries its own 60-line column-major math instead
// (no third-party dependency) and replaces MaterialDescription's MDL code/params by the fixed-layout argument block.
// RenderType::eCompute -- declared but unimplemented in the reference (render.cpp:10-26) -- is the slot HipRender fills.
#pragma once
#include "../../include/strelka_hip.h"

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace oka
{

// ---- tiny column-major math (glm conventions: m[col][row], M * v) ----
struct float3
{
    float x = 0, y = 0, z = 0;
};
struct float4
{
    float x = 0, y = 0, z = 0, w = 0;
};
struct quat
{
    float w = 1, x = 0, y = 0, z = 0;
};
struct float4x4
{
    float m[4][4]; // m[col][row]
    const float* operator[](int col) const // glm's m[col][row]
    {
        return m[col];
    }
    float* operator[](int col)
    {
        return m[col];
    }
    float4x4();
    explicit float4x4(float diag);
    static float4x4 translate(const float3& t);
    static float4x4 scale(const float3& s);
    static float4x4 fromQuat(const quat& q);
    float4x4 operator*(const float4x4& b) const;
    float4 operator*(const float4& v) const;
    float4x4 transposed() const;
    float4x4 inverse() const; // general 4x4, fp64 internally
    bool operator!=(const float4x4& b) const;
};
inline float4x4 inverse(const float4x4& m) // glm::inverse
{
    return m.inverse();
}
quat quatFromEulerRadians(const float3& e); // glm::quat(vec3 eulerAngles)
quat quatFromRotationRows(float r00, float r01, float r02, float r10, float r11, float r12, float r20, float r21, float r22);

// ---- settings/settings.h ----
class SettingsManager
{
public:
    template <typename T>
    void setAs(const char* name, const T& value);
    template <typename T>
    T getAs(const char* name);
    bool has(const char* name) const
    {
        return mMap.find(name) != mMap.end();
    }

private:
    std::unordered_map<std::string, std::string> mMap;
    bool isNameValid(const char* name); // reference: prints "The setting <name> does not exist" and assert(0)
};

class Render;
struct SharedContext // common.h:22-28
{
    size_t mFrameNumber = 0;
    size_t mSubframeIndex = 0;
    SettingsManager* mSettingsManager = nullptr;
    Render* mRender = nullptr;
};
enum class Result : uint32_t
{
    eOk,
    eFail,
    eOutOfMemory
};

// ---- render/buffer.h ----
enum class BufferFormat : char
{
    UNSIGNED_BYTE4,
    FLOAT4,
    FLOAT3
};
struct BufferDesc
{
    uint32_t width;
    uint32_t height;
    BufferFormat format;
};
class Buffer
{
public:
    virtual ~Buffer() = default;
    virtual void resize(uint32_t width, uint32_t height) = 0;
    virtual void* map() = 0;
    virtual void unmap() = 0;
    uint32_t width() const
    {
        return mWidth;
    }
    uint32_t height() const
    {
        return mHeight;
    }
    virtual void* getHostPointer()
    {
        return mHostData.data();
    }
    virtual size_t getHostDataSize()
    {
        return mHostData.size();
    }
    static size_t getElementSize(BufferFormat format);
    size_t getElementSize() const
    {
        return getElementSize(mFormat);
    }
    BufferFormat getFormat() const
    {
        return mFormat;
    }

protected:
    uint32_t mWidth = 0u, mHeight = 0u;
    BufferFormat mFormat = BufferFormat::FLOAT4;
    std::vector<char> mHostData;
};

// ---- scene/camera.h ----
class Camera
{
public:
    std::string name = "Default camera";
    float fov = 45.0f;
    float znear = 0.1f, zfar = 1000.0f;
    quat mOrientation;
    float3 position{ 0.0f, 0.0f, 10.0f };
    struct Matrices
    {
        float4x4 perspective, invPerspective, view;
    } matrices;
    void updateViewMatrix(); // camera.cpp:10-23 (firstperson: rotM * transM)
    void setPerspective(float fov, float aspect, float znear, float zfar); // camera.cpp:125-131 (reverse z)
    void updateAspectRatio(float aspect); // camera.cpp:168-171
    void setPosition(const float3& p);
    void setRotation(const quat& q);
    void lookAt(const float3& eye, const float3& target, const float3& up); // convenience (not in the reference)
};

// ---- scene/scene.h ----
struct Mesh
{
    uint32_t mIndex, mCount, mVbOffset, mVertexCount;
};
struct Curve
{
    enum class Type : uint8_t
    {
        eLinear,
        eCubic
    };
    uint32_t mVertexCountsStart, mVertexCountsCount, mPointsStart, mPointsCount, mWidthsStart, mWidthsCount;
};
struct Instance
{
    float4x4 transform;
    enum class Type : uint8_t
    {
        eMesh,
        eLight,
        eCurve
    } type;
    union
    {
        uint32_t mMeshId;
        uint32_t mCurveId;
    };
    uint32_t mMaterialId = 0;
    uint32_t mLightId = (uint32_t)-1;
};

uint32_t packNormals(const float3& normal); // scene.cpp:111-117
uint32_t packUV(float u, float v); // HdStrelka/RenderPass.cpp:61-67

#ifdef SKH_MIRROR_REFERENCE_MATERIALS
class MaterialManager // (tests only, see Scene::MaterialDescription below: materialmanager.h:33-48)
{
public:
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
};
#endif
class Scene
{
public:
#ifdef SKH_MIRROR_REFERENCE_MATERIALS
    // (tests only: tests/cpp/strelka_lookalike/) the REFERENCE's shape of the description (scene.h:65-78) instead of the argument block, so that the
    // SKH_WITH_STRELKA_HEADERS branches of integration/*.{h,cpp} can be pushed through a compiler (-fsyntax-only) without glm / MDL / OpenUSD
    struct MaterialDescription
    {
        std::string code, file, name;
        bool hasColor = false;
        float3 color;
        std::vector<MaterialManager::Param> params;
    };
#else
    struct MaterialDescription // the MDL code / params of the reference are replaced by the fixed argument block
    {
        std::string name;
        skh_material args;
    };
#endif
    struct Vertex
    {
        float3 pos;
        uint32_t tangent;
        uint32_t normal;
        uint32_t uv;
        float pad0;
        float pad1;
    };
    struct Light // scene.h:146-155 == UniformLight
    {
        float4 points[4];
        float4 color{ 1, 1, 1, 1 };
        float4 normal;
        int type = 0;
        float halfAngle = 0;
        float pad0 = 0, pad1 = 0;
    };
    struct UniformLightDesc // scene.h:157-180
    {
        int32_t type = 0;
        float4x4 xform{ 1.0f };
        float3 position;
        float3 orientation; // euler angles in degrees
        bool useXform = false;
        float3 color{ 1, 1, 1 };
        float intensity = 1.0f;
        float width = 1.0f, height = 1.0f, radius = 0.0f, halfAngle = 0.0f;
    };

    uint32_t createMesh(const std::vector<Vertex>& vb, const std::vector<uint32_t>& ib); // scene.cpp:15-49
    uint32_t createInstance(Instance::Type type, uint32_t geomId, uint32_t materialId, const float4x4& transform,
                            uint32_t lightId = (uint32_t)-1); // scene.cpp:51-87
    uint32_t addMaterial(const MaterialDescription& material); // scene.cpp:89-95
    // RGBA8 image as stbi_load(..., STBI_rgb_alpha) returns it (OptixRender.cpp:1191-1264).  Returns the texture id that
    // skh_material::base_color_texture / normal_texture refer to (1-based; 0 = none).
    struct Texture
    {
        uint32_t width = 0, height = 0;
        std::vector<uint8_t> rgba8;
    };
    uint32_t addTexture(uint32_t width, uint32_t height, const uint8_t* rgba8);
    const std::vector<Texture>& getTextures() const
    {
        return mTextures;
    }
    uint32_t createCurve(Curve::Type type, const std::vector<uint32_t>& vertexCounts, const std::vector<float3>& points,
                         const std::vector<float>& widths);
    uint32_t createLight(const UniformLightDesc& desc); // scene.cpp:306-351
    void updateLight(uint32_t lightId, const UniformLightDesc& desc); // scene.cpp:353-408
    uint32_t addCamera(Camera& camera);
    Camera& getCamera(uint32_t index)
    {
        return mCameras[index];
    }
    float4x4 getTransform(const UniformLightDesc& desc); // scene.h:331-343
    // flat binary dump of the arrays render() uploads (".skscene"; format in strelka_amd/scene_io.py)
    bool saveDump(const std::string& path) const;
    bool loadDump(const std::string& path); // replaces the scene's content; false (scene untouched) on a malformed file
    size_t getCameraCount() // (scene.h:284-288: size_t, not const -- it takes the camera mutex there)
    {
        return mCameras.size();
    }

    std::vector<Vertex>& getVertices()
    {
        return mVertices;
    }
    std::vector<uint32_t>& getIndices()
    {
        return mIndices;
    }
    std::vector<MaterialDescription>& getMaterials()
    {
        return mMaterialsDescs;
    }
    std::vector<Light>& getLights()
    {
        return mLights;
    }
    const std::vector<Instance>& getInstances() const
    {
        return mInstances;
    }
    const std::vector<Mesh>& getMeshes() const
    {
        return mMeshes;
    }
    const std::vector<Curve>& getCurves() const
    {
        return mCurves;
    }
    const std::vector<float3>& getCurvesPoint() const
    {
        return mCurvePoints;
    }
    const std::vector<float>& getCurvesWidths() const
    {
        return mCurveWidths;
    }
    const std::vector<uint32_t>& getCurvesVertexCounts() const
    {
        return mCurveVertexCounts;
    }

private:
    std::vector<Vertex> mVertices;
    std::vector<uint32_t> mIndices;
    std::vector<float3> mCurvePoints;
    std::vector<float> mCurveWidths;
    std::vector<uint32_t> mCurveVertexCounts;
    std::vector<Mesh> mMeshes;
    std::vector<Curve> mCurves;
    std::vector<Instance> mInstances;
    std::vector<Light> mLights;
    std::vector<UniformLightDesc> mLightDesc;
    std::vector<MaterialDescription> mMaterialsDescs;
    std::vector<Texture> mTextures;
    std::vector<Camera> mCameras;
    int mRectLightMeshId = -1, mSphereLightMeshId = -1, mDiskLightMeshId = -1;
    uint32_t createRectLightMesh();
    uint32_t createSphereLightMesh();
    uint32_t createDiscLightMesh();
};

// ---- render/render.h ----
enum class RenderType : int
{
    eOptiX = 0,
    eMetal,
    eCompute
};
class Render
{
public:
    virtual ~Render() = default;
    virtual void init() = 0;
    virtual void render(Buffer* output) = 0;
    virtual Buffer* createBuffer(const BufferDesc& desc) = 0;
    virtual void* getNativeDevicePtr()
    {
        return nullptr;
    }
    void setSharedContext(SharedContext* ctx)
    {
        mSharedCtx = ctx;
    }
    SharedContext& getSharedContext()
    {
        return *mSharedCtx;
    }
    void setScene(Scene* scene)
    {
        mScene = scene;
    }
    Scene* getScene()
    {
        return mScene;
    }

protected:
    SharedContext* mSharedCtx = nullptr;
    Scene* mScene = nullptr;
};
class RenderFactory
{
public:
    static Render* createRender(RenderType type); // eCompute -> HipRender; eOptiX / eMetal -> nullptr here
    static Render* createRender(); // the MI355X backend
};

} // namespace oka
