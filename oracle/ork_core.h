// ORACLE -- TEST INFRASTRUCTURE ONLY (see ork_math.h).
//
// ork_core.h: sampler, camera ray, light sampling/pdfs, vertex attribute packing, offset_ray, curve
// polynomial helpers, accumulation.  Every function cites the reference lines it restates
// (paths relative to arhix52/Strelka).  Pinned against the reference's own headers compiled on the host:
// oracle/ref_golden.cpp -> tests/golden/*.bin -> tests/test_oracle_golden.py.
#pragma once
#include "ork_math.h"

namespace ork
{

// ------------------------------------------------------------------------------------------------
// Sampler  (src/render/optix/RandomSampler.h)
// ------------------------------------------------------------------------------------------------
static const float kOneMinusEps = 0x1.fffffep-1f; // RandomSampler.h:6

enum SampleDimension : uint32_t // RandomSampler.h:13-26
{
    ePixelX = 0,
    ePixelY,
    eLightId,
    eLightPointX,
    eLightPointY,
    eBSDF0,
    eBSDF1,
    eBSDF2,
    eBSDF3,
    eRussianRoulette,
    eNUM_DIMENSIONS
};

struct SamplerState // RandomSampler.h:28-33
{
    uint32_t seed;
    uint32_t sampleIdx;
    uint32_t depth;
};

// Sobol generator matrices for 5 dimensions (RandomSampler.h:139-164).  The table is data from
// Joe & Kuo's direction numbers, regenerated here by rule where possible and spelled out otherwise;
// dimension 0 is the identity (van der Corput), dimension 1 is Pascal's triangle mod 2.
extern uint32_t kSobolMatrix[5][32];

static inline uint32_t hash_murmur(uint32_t x) // RandomSampler.h:86-95 (murmur3 finalizer)
{
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
static inline uint32_t hash_combine(uint32_t seed, uint32_t v) // RandomSampler.h:50-53
{
    return seed ^ (v + (seed << 6) + (seed >> 2));
}
static inline uint32_t part1by1(uint32_t x) // RandomSampler.h:115-123
{
    x &= 0x0000ffffu;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    x = (x ^ (x << 1)) & 0x55555555u;
    return x;
}
static inline uint32_t encode_morton2(uint32_t x, uint32_t y) // RandomSampler.h:125-128
{
    return (part1by1(y) << 1) + part1by1(x);
}
static inline SamplerState init_sampler(uint32_t px, uint32_t py, uint32_t pixelSampleIndex, uint32_t maxSampleCount,
                                        uint32_t seed) // RandomSampler.h:130-137
{
    SamplerState s;
    s.seed = seed;
    s.sampleIdx = encode_morton2(px, py) * maxSampleCount + pixelSampleIndex;
    s.depth = 0;
    return s;
}
static inline uint32_t sobol_uint(uint32_t index, uint32_t dim) // RandomSampler.h:166-175
{
    uint32_t X = 0;
    for (int bit = 0; bit < 32; ++bit)
    {
        const uint32_t mask = (index >> bit) & 1u;
        X ^= mask * kSobolMatrix[dim][bit];
    }
    return X;
}
static inline uint32_t laine_karras_permutation(uint32_t value, uint32_t seed) // RandomSampler.h:182-190
{
    value += seed;
    value ^= value * 0x6c50b47cu;
    value ^= value * 0xb82f1e52u;
    value ^= value * 0xc7afe638u;
    value ^= value * 0x8d22f6e6u;
    return value;
}
static inline uint32_t reverse_bits(uint32_t v) // RandomSampler.h:192-203
{
    v = ((v & 0xaaaaaaaau) >> 1) | ((v & 0x55555555u) << 1);
    v = ((v & 0xccccccccu) >> 2) | ((v & 0x33333333u) << 2);
    v = ((v & 0xf0f0f0f0u) >> 4) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v & 0xff00ff00u) >> 8) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
}
static inline uint32_t nested_uniform_scramble(uint32_t value, uint32_t seed) // RandomSampler.h:205-211
{
    value = reverse_bits(value);
    value = laine_karras_permutation(value, seed);
    value = reverse_bits(value);
    return value;
}
static inline uint32_t sobol_scramble_uint(uint32_t index, uint32_t dim, uint32_t seed) // RandomSampler.h:213-219
{
    seed = hash_murmur(seed);
    index = nested_uniform_scramble(index, seed);
    return nested_uniform_scramble(sobol_uint(index, dim), hash_combine(seed, dim));
}
static inline float sobol_scramble(uint32_t index, uint32_t dim, uint32_t seed)
{
    const uint32_t r = sobol_scramble_uint(index, dim, seed);
    return fminf((float)r * 0x1p-32f, kOneMinusEps); // uint->float conversion rounds to nearest even
}
// random<Dim>(state): RandomSampler.h:221-226.  Note the % 5: the 10 logical dimensions alias pairwise.
static inline float sampler_random(const SamplerState& s, uint32_t dim)
{
    const uint32_t dimension = (dim + s.depth * (uint32_t)eNUM_DIMENSIONS) % 5u;
    return sobol_scramble(s.sampleIdx, dimension, s.seed + s.depth);
}

// ------------------------------------------------------------------------------------------------
// Camera ray  (src/render/optix/OptixRender.cu:38-58)
// ------------------------------------------------------------------------------------------------
static inline void generate_camera_ray(uint32_t px, uint32_t py, uint32_t width, uint32_t height, const float* clipToView,
                                       const float* viewToWorld, float jx, float jy, f3& origin, f3& direction)
{
    const float posx = (float)px + jx;
    const float posy = (float)py + jy;
    // float2 / float2 is component-wise true division (sutil/vec_math.h float2 operator/)
    const float ndcx = (posx / (float)width) * 2.0f - 1.0f;
    const float ndcy = (posy / (float)height) * 2.0f - 1.0f;
    const f4 clip{ ndcx, ndcy, 1.0f, 1.0f };
    const f4 viewSpace = mul44(clipToView, clip);
    const f4 wdir = mul44(viewToWorld, f4{ viewSpace.x, viewSpace.y, viewSpace.z, 0.0f });
    origin = mk3(mul44(viewToWorld, f4{ 0.0f, 0.0f, 0.0f, 1.0f }));
    direction = normalize(mk3(wdir));
}

// Reverse-Z projection inverse the reference builds by hand (src/scene/camera.cpp:61-131), returned as the
// row-major clipToView that render() uploads (OptixRender.cpp:954: transpose of the glm column-major matrix).
// setPerspective swaps near/far: perspective(fov, aspect, zfar, znear).
static inline void make_clip_to_view(float fovDeg, float aspect, float znear, float zfar, float* out16)
{
    const float n = zfar, f = znear; // swapped for reverse z (camera.cpp:125-131)
    const float radians = fovDeg * 0.01745329251994329576923690768489f; // glm::radians
    const float focal_length = 1.0f / tanf(radians / 2.0f);
    const float x = focal_length / aspect;
    const float y = focal_length;
    const float A = n / (f - n);
    const float B = f * A;
    const float m[16] = { 1 / x, 0, 0, 0, 0, 1 / y, 0, 0, 0, 0, 0, -1.0f, 0, 0, 1 / B, A / B };
    for (int i = 0; i < 16; ++i)
        out16[i] = m[i];
}

// ------------------------------------------------------------------------------------------------
// Lights  (include/render/Lights.h)
// ------------------------------------------------------------------------------------------------
static const float kPi = 3.14159265358979323846f; // M_PIf (sutil/vec_math.h:43-44)

struct UniformLight // Lights.h:5-14, 112 B
{
    f4 points[4];
    f4 color;
    f4 normal;
    int32_t type;
    float halfAngle;
    float pad0, pad1;
};
static_assert(sizeof(UniformLight) == 112, "UniformLight layout");

struct LightSampleData // Lights.h:16-26
{
    f3 pointOnLight;
    float pdf;
    f3 normal;
    float area;
    f3 L;
    float distToLight;
};

static inline float mis_weight_balance(float a, float b) // Lights.h:28-31
{
    return 1.0f / (1.0f + (b / a));
}
static inline float calc_light_area(const UniformLight& l) // Lights.h:33-52
{
    float area = 0.0f;
    if (l.type == 0)
    {
        const f3 e1 = mk3(l.points[1]) - mk3(l.points[0]);
        const f3 e2 = mk3(l.points[3]) - mk3(l.points[0]);
        area = length(cross(e1, e2));
    }
    else if (l.type == 1)
        area = kPi * l.points[0].x * l.points[0].x;
    else if (l.type == 2)
        area = 4.0f * kPi * l.points[0].x * l.points[0].x;
    return area;
}
static inline f3 calc_light_normal(const UniformLight& l, const f3& hitPoint) // Lights.h:54-74
{
    f3 norm = mk3(0.0f);
    if (l.type == 0)
    {
        const f3 e1 = mk3(l.points[1]) - mk3(l.points[0]);
        const f3 e2 = mk3(l.points[3]) - mk3(l.points[0]);
        norm = -normalize(cross(e1, e2));
    }
    else if (l.type == 1)
        norm = mk3(l.normal);
    else if (l.type == 2)
        norm = normalize(hitPoint - mk3(l.points[1]));
    return norm;
}
static inline void fill_light_data(const UniformLight& l, const f3& hitPoint, LightSampleData& d) // Lights.h:76-84
{
    d.area = calc_light_area(l);
    d.normal = calc_light_normal(l, hitPoint);
    const f3 toLight = d.pointOnLight - hitPoint;
    const float lenToLight = length(toLight);
    d.L = toLight / lenToLight;
    d.distToLight = lenToLight;
}

struct SphQuad // Lights.h:86-94
{
    f3 o, x, y, z;
    float z0, z0sq;
    float x0, y0, y0sq;
    float x1, y1, y1sq;
    float b0, b1, b0sq, k;
    float S;
};
static inline SphQuad sph_quad_init(const UniformLight& l, const f3& o) // Lights.h:97-153
{
    SphQuad q;
    const f3 ex = mk3(l.points[1]) - mk3(l.points[0]);
    const f3 ey = mk3(l.points[3]) - mk3(l.points[0]);
    const f3 s = mk3(l.points[0]);
    const float exl = length(ex);
    const float eyl = length(ey);
    q.o = o;
    q.x = ex / exl;
    q.y = ey / eyl;
    q.z = cross(q.x, q.y);
    const f3 d = s - o;
    q.z0 = dot(d, q.z);
    if (q.z0 > 0)
    {
        q.z = q.z * -1.0f;
        q.z0 *= -1.0f;
    }
    q.z0sq = q.z0 * q.z0;
    q.x0 = dot(d, q.x);
    q.y0 = dot(d, q.y);
    q.x1 = q.x0 + exl;
    q.y1 = q.y0 + eyl;
    q.y0sq = q.y0 * q.y0;
    q.y1sq = q.y1 * q.y1;
    const f3 v00{ q.x0, q.y0, q.z0 };
    const f3 v01{ q.x0, q.y1, q.z0 };
    const f3 v10{ q.x1, q.y0, q.z0 };
    const f3 v11{ q.x1, q.y1, q.z0 };
    const f3 n0 = normalize(cross(v00, v10));
    const f3 n1 = normalize(cross(v10, v11));
    const f3 n2 = normalize(cross(v11, v01));
    const f3 n3 = normalize(cross(v01, v00));
    const float g0 = skm::acosf_(-dot(n0, n1));
    const float g1 = skm::acosf_(-dot(n1, n2));
    const float g2 = skm::acosf_(-dot(n2, n3));
    const float g3 = skm::acosf_(-dot(n3, n0));
    q.b0 = n0.z;
    q.b1 = n2.z;
    q.b0sq = q.b0 * q.b0;
    q.k = 2.0f * kPi - g2 - g3;
    q.S = g0 + g1 - q.k;
    return q;
}
static inline f3 sph_quad_sample(const SphQuad& q, float u, float v) // Lights.h:155-189
{
    const float au = u * q.S + q.k;
    const float fu = (skm::cosf_(au) * q.b0 - q.b1) / skm::sinf_(au);
    float cu = 1.0f / sqrtf(fu * fu + q.b0sq) * (fu > 0.0f ? 1.0f : -1.0f);
    cu = clampf(cu, -1.0f, 1.0f);
    float xu = -(cu * q.z0) / sqrtf(1.0f - cu * cu);
    xu = clampf(xu, q.x0, q.x1);
    const float d = sqrtf(xu * xu + q.z0sq);
    const float h0 = q.y0 / sqrtf(d * d + q.y0sq);
    const float h1 = q.y1 / sqrtf(d * d + q.y1sq);
    const float hv = h0 + v * (h1 - h0);
    const float hv2 = hv * hv;
    const float eps = 1e-5f;
    const float yv = (hv < 1.0f - eps) ? (hv * d) / sqrtf(1 - hv2) : q.y1;
    return (q.o + xu * q.x + yv * q.y + q.z0 * q.z);
}
static inline float rect_light_solid_angle_pdf(const UniformLight& l, const f3& hitPoint) // Lights.h:191-199
{
    const SphQuad q = sph_quad_init(l, hitPoint);
    if (q.S <= 0.0f)
        return 0.0f;
    return 1.0f / q.S;
}
static inline float get_rect_light_pdf(const UniformLight& l, const f3& lightHitPoint,
                                       const f3& surfaceHitPoint) // Lights.h:201-209
{
    LightSampleData d{};
    d.pointOnLight = lightHitPoint;
    fill_light_data(l, surfaceHitPoint, d);
    d.pdf = d.distToLight * d.distToLight / (dot(-d.L, d.normal) * d.area);
    return d.pdf;
}
static inline float get_direct_light_pdf(float angle) // Lights.h:211-214: float * float(1 - cosf)
{
    return 1.0f / (2.0f * kPi * (1.0f - skm::cosf_(angle)));
}
static inline float get_sphere_light_pdf() // Lights.h:216-219
{
    return 1.0f / (4.0f * kPi);
}
static inline float get_light_pdf(const UniformLight& l, const f3& lightHitPoint,
                                  const f3& surfaceHitPoint) // Lights.h:221-243
{
    switch (l.type)
    {
    case 0:
        return get_rect_light_pdf(l, lightHitPoint, surfaceHitPoint);
    case 2:
        return get_sphere_light_pdf();
    case 3:
        return get_direct_light_pdf(l.halfAngle);
    default:
        break;
    }
    return 0.0f; // disc (type 1) has no pdf in the reference
}
static inline LightSampleData sample_rect_light(const UniformLight& l, float ux, float uy,
                                                const f3& hitPoint) // Lights.h:245-275
{
    LightSampleData d;
    const f3 e1 = mk3(l.points[1]) - mk3(l.points[0]);
    const f3 e2 = mk3(l.points[3]) - mk3(l.points[0]);
    const SphQuad quad = sph_quad_init(l, hitPoint);
    if (quad.S <= 0.0f)
    {
        d.pdf = 0.0f;
        d.pointOnLight = mk3(l.points[0]) + e1 * ux + e2 * uy;
        fill_light_data(l, hitPoint, d);
        return d;
    }
    if (quad.S < 1e-3f)
    {
        d.pointOnLight = mk3(l.points[0]) + e1 * ux + e2 * uy;
        fill_light_data(l, hitPoint, d);
        d.pdf = d.distToLight * d.distToLight / (-dot(d.L, d.normal) * d.area);
        return d;
    }
    d.pointOnLight = sph_quad_sample(quad, ux, uy);
    fill_light_data(l, hitPoint, d);
    d.pdf = 1.0f / quad.S;
    return d;
}
static inline LightSampleData sample_rect_light_uniform(const UniformLight& l, float ux, float uy,
                                                        const f3& hitPoint) // Lights.h:277-289
{
    LightSampleData d;
    const f3 e1 = mk3(l.points[1]) - mk3(l.points[0]);
    const f3 e2 = mk3(l.points[3]) - mk3(l.points[0]);
    d.pointOnLight = mk3(l.points[0]) + e1 * ux + e2 * uy;
    fill_light_data(l, hitPoint, d);
    d.pdf = d.distToLight * d.distToLight / (-dot(d.L, d.normal) * d.area);
    return d;
}
static inline void create_coordinate_system(const f3& N, f3& Nt, f3& Nb) // Lights.h:291-300
{
    if (fabsf(N.x) > fabsf(N.y))
    {
        const float invLen = 1.0f / sqrtf(N.x * N.x + N.z * N.z);
        Nt = f3{ -N.z * invLen, 0.0f, N.x * invLen };
    }
    else
    {
        const float invLen = 1.0f / sqrtf(N.y * N.y + N.z * N.z);
        Nt = f3{ 0.0f, N.z * invLen, -N.y * invLen };
    }
    Nb = cross(N, Nt);
}
// Lights.h:302-317.  The reference mixes double literals into this function; on the device cos/sin/sqrt of a
// float argument resolve to the float overloads, and the double literals promote the surrounding arithmetic:
//   phi      = (float)(2.0 * (double)M_PIf * (double)uv.x)
//   cosTheta = (float)(1.0 - (double)uv.y * (1.0 - (double)cosf(angle)))
//   sinTheta = (float)sqrt(1.0 - (double)(cosTheta*cosTheta))   [cosTheta*cosTheta is float*float; sqrt(double)]
//   pdf      = (float)(1.0 / (2.0 * (double)M_PIf * (1.0 - (double)cosf(angle))))
static inline f3 sample_cone(float ux, float uy, float angle, const f3& direction, float& pdf)
{
    const float phi = (float)(2.0 * (double)kPi * (double)ux);
    const float cosTheta = (float)(1.0 - (double)uy * (1.0 - (double)skm::cosf_(angle)));
    const float sinTheta = (float)sqrt(1.0 - (double)(cosTheta * cosTheta));
    f3 u, v;
    create_coordinate_system(direction, u, v);
    const f3 sampledDir = normalize(skm::cosf_(phi) * sinTheta * u + skm::sinf_(phi) * sinTheta * v + cosTheta * direction);
    pdf = (float)(1.0 / (2.0 * (double)kPi * (1.0 - (double)skm::cosf_(angle))));
    return sampledDir;
}
static inline LightSampleData sample_distant_light(const UniformLight& l, float ux, float uy,
                                                   const f3& hitPoint) // Lights.h:319-333
{
    (void)hitPoint;
    LightSampleData d;
    float pdf = 0.0f;
    const f3 coneSample = sample_cone(ux, uy, l.halfAngle, -mk3(l.normal), pdf);
    d.area = 0.0f;
    d.distToLight = 1e9f;
    d.L = coneSample;
    d.normal = mk3(l.normal);
    d.pdf = pdf;
    d.pointOnLight = coneSample;
    return d;
}
static inline LightSampleData sample_sphere_light(const UniformLight& l, float ux, float uy,
                                                  const f3& hitPoint) // Lights.h:335-362
{
    LightSampleData d;
    const float cosTheta = 1.0f - 2.0f * ux;
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    const float phi = 2.0f * kPi * uy;
    const float radius = l.points[0].x;
    const f3 sphereDirection{ sinTheta * skm::cosf_(phi), sinTheta * skm::sinf_(phi), cosTheta };
    const f3 lightPoint = mk3(l.points[1]) + radius * sphereDirection;
    d.L = normalize(lightPoint - hitPoint);
    d.distToLight = length(lightPoint - hitPoint);
    d.area = 0.0f;
    d.normal = sphereDirection;
    d.pdf = 1.0f / (4.0f * kPi);
    d.pointOnLight = lightPoint;
    return d;
}

// ------------------------------------------------------------------------------------------------
// Vertex attribute packing  (scene.cpp:111-117, HdStrelka/RenderPass.cpp:53-67, closest_hit.cu:236-254)
// ------------------------------------------------------------------------------------------------
static inline uint32_t pack_normal(const f3& n) // scene.cpp:111-117 == RenderPass.cpp:53-59
{
    uint32_t packed = (uint32_t)((n.x + 1.0f) / 2.0f * 511.99999f);
    packed += (uint32_t)((n.y + 1.0f) / 2.0f * 511.99999f) << 10;
    packed += (uint32_t)((n.z + 1.0f) / 2.0f * 511.99999f) << 20;
    return packed;
}
static inline uint32_t pack_uv(float u, float v) // RenderPass.cpp:61-67: range [-10,10], 16-16 bits
{
    uint32_t packed = (uint32_t)((u + 10.0f) / 20.0f * 16383.99999f);
    packed += (uint32_t)((v + 10.0f) / 20.0f * 16383.99999f) << 16;
    return packed;
}
static inline f3 unpack_normal(uint32_t val) // closest_hit.cu:236-244 (z mask is 0xfff00000: 12 bits)
{
    f3 n;
    n.z = (float)((val & 0xfff00000u) >> 20) / 511.99999f * 2.0f - 1.0f;
    n.y = (float)((val & 0x000ffc00u) >> 10) / 511.99999f * 2.0f - 1.0f;
    n.x = (float)(val & 0x000003ffu) / 511.99999f * 2.0f - 1.0f;
    return n;
}
static inline f2 unpack_uv(uint32_t val) // closest_hit.cu:247-254
{
    f2 uv;
    uv.y = (float)((val & 0xffff0000u) >> 16) / 16383.99999f * 20.0f - 10.0f;
    uv.x = (float)(val & 0x0000ffffu) / 16383.99999f * 20.0f - 10.0f;
    return uv;
}

// ------------------------------------------------------------------------------------------------
// 2-D texture lookup.  The reference samples through a CUDA texture object created by loadTextureFromFile
// (OptixRender.cpp:1191-1264): uchar4 array, cudaReadModeNormalizedFloat (byte / 255), cudaFilterModeLinear,
// cudaAddressModeWrap, normalized coordinates; tex_lookup_float4_2d (texture_support_cuda.h:287-313) passes the
// coordinate through unchanged for the default wrap_repeat / crop (0,1) that OmniPBR uses.  The filter itself is
// hardware; restated here from the CUDA C Programming Guide, "Texture Fetching / Linear Filtering":
//   x = N * frac(u);  xB = x - 0.5;  i = floor(xB);  alpha = frac(xB) kept in 1.8 fixed point;
//   tex = (1-a)(1-b) T[i,j] + a(1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1],  indices wrapped modulo N
// (8 fractional bits, round to nearest -- the guide does not state the rounding; parity vs. NVIDIA hardware is unpinned,
// GPU == oracle is exact).  Row 0 of the image is v = 0 (stbi_load order).
// ------------------------------------------------------------------------------------------------
struct OTexture
{
    uint32_t offset, width, height, pad; // offset in texels into the shared RGBA8 array
};
static inline void tex_axis(float u, uint32_t n, uint32_t& i0, uint32_t& i1, float& a)
{
    const float x = (u - floorf(u)) * (float)n - 0.5f;
    const float fl = floorf(x);
    a = floorf((x - fl) * 256.0f + 0.5f) * (1.0f / 256.0f);
    int i = (int)fl; // -1 .. n-1  (frac() == 1.0f cannot happen: u - floorf(u) < 1 in fp32 except for tiny negative u, handled by the modulo)
    i = i < 0 ? i + (int)n : i;
    i0 = (uint32_t)i % n;
    i1 = (i0 + 1u) % n;
}
static inline f4 texel_rgba8(uint32_t t)
{
    return f4{ (float)(t & 0xffu) / 255.0f, (float)((t >> 8) & 0xffu) / 255.0f, (float)((t >> 16) & 0xffu) / 255.0f,
               (float)(t >> 24) / 255.0f };
}
static inline f4 tex_lookup_rgba8(const uint32_t* texels, const OTexture& t, float u, float v)
{
    uint32_t x0, x1, y0, y1;
    float a, b;
    tex_axis(u, t.width, x0, x1, a);
    tex_axis(v, t.height, y0, y1, b);
    const uint32_t* base = texels + t.offset;
    const f4 t00 = texel_rgba8(base[y0 * t.width + x0]), t10 = texel_rgba8(base[y0 * t.width + x1]);
    const f4 t01 = texel_rgba8(base[y1 * t.width + x0]), t11 = texel_rgba8(base[y1 * t.width + x1]);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return f4{ ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x, ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y,
               ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z, ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w };
}

// offset_ray: closest_hit.cu:218-233 (Waechter & Binder, Ray Tracing Gems ch. 6)
static inline f3 offset_ray(const f3& p, const f3& n)
{
    const float origin = 1.0f / 32.0f;
    const float float_scale = 1.0f / 65536.0f;
    const float int_scale = 256.0f;
    const int32_t ofx = (int32_t)(int_scale * n.x);
    const int32_t ofy = (int32_t)(int_scale * n.y);
    const int32_t ofz = (int32_t)(int_scale * n.z);
    const f3 p_i{ i2f(f2i(p.x) + ((p.x < 0) ? -ofx : ofx)), i2f(f2i(p.y) + ((p.y < 0) ? -ofy : ofy)),
                  i2f(f2i(p.z) + ((p.z < 0) ? -ofz : ofz)) };
    return f3{ fabsf(p.x) < origin ? p.x + float_scale * n.x : p_i.x,
               fabsf(p.y) < origin ? p.y + float_scale * n.y : p_i.y,
               fabsf(p.z) < origin ? p.z + float_scale * n.z : p_i.z };
}

static inline f3 interpolate_attrib(const f3& a1, const f3& a2, const f3& a3, float bx,
                                    float by) // closest_hit.cu:199-205
{
    return a1 * (1.0f - bx - by) + a2 * bx + a3 * by;
}

// ------------------------------------------------------------------------------------------------
// Cubic B-spline segment as a polynomial  (cuda/curve.h:170-275, 306-353, 412-417)
// ------------------------------------------------------------------------------------------------
struct CubicInterpolator
{
    f4 p[4];
    void initializeFromBSpline(const f4* q) // curve.h:177-187
    {
        p[0] = (q[0] * (-1.0f) + q[1] * (3.0f) + q[2] * (-3.0f) + q[3]) / 6.0f;
        p[1] = (q[0] * (3.0f) + q[1] * (-6.0f) + q[2] * (3.0f)) / 6.0f;
        p[2] = (q[0] * (-3.0f) + q[2] * (3.0f)) / 6.0f;
        p[3] = (q[0] * (1.0f) + q[1] * (4.0f) + q[2] * (1.0f)) / 6.0f;
    }
    f4 position4(float u) const // curve.h:237-240
    {
        return (((p[0] * u) + p[1]) * u + p[2]) * u + p[3];
    }
    f4 velocity4(float u) const // curve.h:252-260
    {
        if (u == 0)
            u = 0.000001f;
        if (u == 1)
            u = 0.999999f;
        return ((3.0f * p[0] * u) + 2.0f * p[1]) * u + p[2];
    }
    f4 acceleration4(float u) const // curve.h:272-275
    {
        return 6.0f * p[0] * u + 2.0f * p[1];
    }
};
// surfaceNormal<CubicInterpolator, 2>: curve.h:306-353.  ps is moved onto the surface.
static inline f3 curve_surface_normal(const CubicInterpolator& bc, float u, f3& ps)
{
    f3 normal;
    if (u == 0.0f)
        normal = -mk3(bc.velocity4(0));
    else if (u == 1.0f)
        normal = mk3(bc.velocity4(1));
    else
    {
        const f4 p4 = bc.position4(u);
        const f3 p = mk3(p4);
        const float r = p4.w;
        const f4 d4 = bc.velocity4(u);
        const f3 d = mk3(d4);
        const float dr = d4.w;
        float dd = dot(d, d);
        f3 o1 = ps - p;
        o1 -= (dot(o1, d) / dd) * d;
        o1 *= r / length(o1);
        ps = p + o1;
        dd -= dot(mk3(bc.acceleration4(u)), o1);
        normal = dd * o1 - (dr * r) * d;
    }
    return normalize(normal);
}
static inline f3 curve_tangent(const CubicInterpolator& bc, float u) // curve.h:412-417
{
    return normalize(mk3(bc.velocity4(u)));
}

// ------------------------------------------------------------------------------------------------
// Accumulation  (postprocessing/Utils.h:5-14, OptixRender.cu:60-78)
// ------------------------------------------------------------------------------------------------
static inline f3 tonemap(f3 color, const f3& exposure) // Utils.h:5-9
{
    color *= exposure;
    return color / (color + mk3(1.0f));
}
static inline f3 inverse_tonemap(const f3& color, const f3& exposure) // Utils.h:11-14
{
    return color / (exposure - color * exposure);
}
// returns the new accumulated colour; history is updated by the caller (OptixRender.cu:60-78)
static inline f3 accumulate(const f3& prev, const f3& value, const f3& exposure, uint32_t subFrameIndex)
{
    f3 accumColor = value;
    if (subFrameIndex > 0)
    {
        const float a = 1.0f / (float)(subFrameIndex + 1);
        accumColor = inverse_tonemap(lerp3(tonemap(prev, exposure), tonemap(accumColor, exposure), a), exposure);
    }
    return accumColor;
}
// exposure from the photographic settings (OptixRender.cpp:961-987)
static inline void compute_exposure(float filmIso, float cm2_factor, float fStop, float shutterSpeed, float* out3)
{
    const f3 whitePoint{ 1.0f, 1.0f, 1.0f };
    f3 e = all3(whitePoint) ? f3{ 1.0f / whitePoint.x, 1.0f / whitePoint.y, 1.0f / whitePoint.z } : mk3(1.0f);
    const float lum = dot(e, f3{ 0.299f, 0.587f, 0.114f });
    if (filmIso > 0.0f)
        e *= cm2_factor * filmIso / (shutterSpeed * fStop * fStop) / 100.0f;
    else
        e *= cm2_factor;
    e = e / lum;
    out3[0] = e.x;
    out3[1] = e.y;
    out3[2] = e.z;
}

// Tonemappers (postprocessing/Tonemappers.cu:6-109)
static inline f3 tm_reinhard(const f3& c)
{
    const float lum = dot(c, f3{ 0.299f, 0.587f, 0.114f });
    return c / (lum + 1);
}
static inline f3 tm_aces_film(const f3& x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    const f3 r = (x * (a * x + mk3(b))) / (x * (c * x + mk3(d)) + mk3(e));
    return f3{ saturatef(r.x), saturatef(r.y), saturatef(r.z) };
}
static inline f3 tm_aces_fitted(f3 color)
{
    // matrices are written with double literals in the reference and stored into float Matrix3x3
    const float in[9] = { 0.59719f, 0.35458f, 0.04823f, 0.07600f, 0.90834f, 0.01566f, 0.02840f, 0.13383f, 0.83777f };
    const float out[9] = { 1.60475f, -0.53108f, -0.07367f, -0.10208f, 1.10813f, -0.00605f,
                           -0.00327f, -0.07276f, 1.07602f };
    f3 c{ in[0] * color.x + in[1] * color.y + in[2] * color.z, in[3] * color.x + in[4] * color.y + in[5] * color.z,
          in[6] * color.x + in[7] * color.y + in[8] * color.z };
    const f3 a = c * (c + mk3(0.0245786f)) - mk3(0.000090537f);
    const f3 b = c * (0.983729f * c + mk3(0.4329510f)) + mk3(0.238081f);
    c = a / b;
    const f3 o{ out[0] * c.x + out[1] * c.y + out[2] * c.z, out[3] * c.x + out[4] * c.y + out[5] * c.z,
                out[6] * c.x + out[7] * c.y + out[8] * c.z };
    return f3{ saturatef(o.x), saturatef(o.y), saturatef(o.z) };
}

} // namespace ork
