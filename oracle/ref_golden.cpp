// ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden-vector generator.
//
// This translation unit #includes the reference's own host-compilable headers FROM WHERE THEY LIE under
// /root/reference (never copied into this repository) and writes small binary fixtures:
//   src/render/optix/RandomSampler.h, include/render/Lights.h, src/render/optix/postprocessing/Utils.h,
//   sutil/vec_math.h, sutil/vec_math_adv.h
// Build recipe: oracle/Makefile target `ref` (CUDA decorators are defined away with -D on the command line;
// vector_types.h comes from the CUDA headers bundled with this image's Triton wheel).
// Run by tests/golden/make_golden.py, which commits the outputs under tests/golden/.
// The inputs of every case are generated here (fixed lattices / an LCG) and stored next to the outputs, so the
// tests need nothing from /root/reference at run time.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <math.h>
#include <string>
#include <vector>
using std::isnan;
using std::max;
using std::min;

#include <vector_functions.h>
#include <vector_types.h>
#include <sutil/vec_math.h>
#include <sutil/vec_math_adv.h>

#include <RandomSampler.h>
#include <Lights.h>
#include <postprocessing/Utils.h>

static std::string g_dir;
template <typename T>
static void dump(const char* name, const std::vector<T>& v)
{
    const std::string p = g_dir + "/" + name;
    FILE* f = fopen(p.c_str(), "wb");
    if (!f)
    {
        perror(p.c_str());
        exit(1);
    }
    fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
    printf("%s: %zu x %zu B\n", name, v.size(), sizeof(T));
}
static uint32_t g_lcg = 12345u;
static float frand()
{
    g_lcg = g_lcg * 1664525u + 1013904223u;
    return (float)(g_lcg >> 8) * (1.0f / 16777216.0f);
}

template <int D>
static float rnd(SamplerState& s)
{
    return random<(SampleDimension)D>(s);
}
static float rnd_dim(SamplerState& s, int d)
{
    switch (d)
    {
    case 0:
        return rnd<0>(s);
    case 1:
        return rnd<1>(s);
    case 2:
        return rnd<2>(s);
    case 3:
        return rnd<3>(s);
    case 4:
        return rnd<4>(s);
    case 5:
        return rnd<5>(s);
    case 6:
        return rnd<6>(s);
    case 7:
        return rnd<7>(s);
    case 8:
        return rnd<8>(s);
    default:
        return rnd<9>(s);
    }
}

static void put3(std::vector<float>& v, const float3& a)
{
    v.push_back(a.x);
    v.push_back(a.y);
    v.push_back(a.z);
}
static void put_lsd(std::vector<float>& v, const LightSampleData& d)
{
    put3(v, d.pointOnLight);
    v.push_back(d.pdf);
    put3(v, d.normal);
    v.push_back(d.area);
    put3(v, d.L);
    v.push_back(d.distToLight);
}

int main(int argc, char** argv)
{
    g_dir = argc > 1 ? argv[1] : ".";
    // ---- 1. sampler: inputs (x, y, sampleIndex, depth, dim) uint32 x5 per case, outputs float + sampleIdx ----
    {
        const uint32_t xs[] = { 0, 1, 3, 100, 1919 }, ys[] = { 0, 2, 5, 200, 1079 }, ss[] = { 0, 1, 7, 63 };
        std::vector<uint32_t> in;
        std::vector<float> out;
        std::vector<uint32_t> idx;
        for (uint32_t x : xs)
            for (uint32_t y : ys)
                for (uint32_t si : ss)
                    for (uint32_t depth = 0; depth < 6; ++depth)
                        for (int d = 0; d < 10; ++d)
                        {
                            SamplerState s = initSampler(x, y, y * 1920 + x, si, 64, 52u);
                            idx.push_back(s.sampleIdx);
                            s.depth = depth;
                            in.insert(in.end(), { x, y, si, depth, (uint32_t)d });
                            out.push_back(rnd_dim(s, d));
                        }
        dump("sampler_in.u32", in);
        dump("sampler_out.f32", out);
        dump("sampler_idx.u32", idx);
        // the 4K / 256 spp corner (largest index in any BASELINE config, no uint32 wrap)
        std::vector<uint32_t> big;
        big.push_back(initSampler(3839, 2159, 0, 255, 256, 52u).sampleIdx);
        big.push_back(EncodeMorton2(3839, 2159));
        dump("sampler_big.u32", big);
        // raw Sobol words and the table itself (data)
        std::vector<uint32_t> sob;
        for (uint32_t d = 0; d < 5; ++d)
            for (uint32_t i = 0; i < 64; ++i)
                sob.push_back(sobol_uint(i * 2654435761u + d, d));
        dump("sobol_uint.u32", sob);
        std::vector<uint32_t> tab(&sb_matrix[0][0], &sb_matrix[0][0] + 160);
        dump("sobol_matrix.u32", tab);
    }
    // ---- 2. lights ----
    {
        UniformLight rect{};
        // rect light built as Scene::updateLight does for width 0.6, height 0.4, at z = 2 facing -z (scene.cpp:356-369)
        rect.points[0] = make_float4(0.3f, 0.2f, 2.0f, 1.0f);
        rect.points[1] = make_float4(-0.3f, 0.2f, 2.0f, 1.0f);
        rect.points[2] = make_float4(-0.3f, -0.2f, 2.0f, 1.0f);
        rect.points[3] = make_float4(0.3f, -0.2f, 2.0f, 1.0f);
        rect.color = make_float4(17.0f, 12.0f, 4.0f, 1.0f);
        rect.type = 0;
        UniformLight sph{};
        sph.points[0] = make_float4(0.25f, 0, 0, 0);
        sph.points[1] = make_float4(0.5f, 1.5f, -0.25f, 1.0f);
        sph.color = make_float4(5.0f, 5.0f, 5.0f, 1.0f);
        sph.type = 2;
        UniformLight dist{};
        dist.type = 3;
        dist.halfAngle = 0.0872664626f; // 5 degrees
        dist.normal = make_float4(normalize(make_float3(0.3f, -0.9f, 0.2f)), 0.0f);
        dist.color = make_float4(2.0f, 2.0f, 2.0f, 1.0f);
        std::vector<float> lights;
        for (const UniformLight* l : { &rect, &sph, &dist })
        {
            const float* p = reinterpret_cast<const float*>(l);
            lights.insert(lights.end(), p, p + sizeof(UniformLight) / 4);
        }
        dump("lights_def.f32", lights); // 3 x 28 words (type is an int in word 24)

        std::vector<float> in, oU, oS, oSph, oD, oPdf, oNrm;
        // hit points: a lattice below the light, some far away (S < 1e-3), some behind the light plane
        std::vector<float3> P;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                P.push_back(make_float3(-0.8f + 0.5f * i, -0.7f + 0.45f * j, 0.1f * (i + j)));
        P.push_back(make_float3(0.0f, 0.0f, 0.0f));
        P.push_back(make_float3(30.0f, 5.0f, -40.0f)); // far: S < 1e-3 branch
        P.push_back(make_float3(100.0f, -80.0f, -300.0f)); // farther
        P.push_back(make_float3(0.1f, 0.1f, 3.0f)); // behind the emitting side
        for (const float3& p : P)
            for (int k = 0; k < 6; ++k)
            {
                const float2 u = make_float2(frand(), frand());
                in.insert(in.end(), { p.x, p.y, p.z, u.x, u.y });
                put_lsd(oU, SampleRectLightUniform(rect, u, p));
                put_lsd(oS, SampleRectLight(rect, u, p));
                put_lsd(oSph, SampleSphereLight(sph, u, p));
                put_lsd(oD, SampleDistantLight(dist, u, p));
                // pdf queries: light hit point = the uniform sample
                const LightSampleData d = SampleRectLightUniform(rect, u, p);
                oPdf.push_back(getLightPdf(rect, d.pointOnLight, p));
                oPdf.push_back(getLightPdf(rect, p)); // solid-angle variant (Lights.h:191-199)
                oPdf.push_back(getLightPdf(sph, d.pointOnLight, p));
                oPdf.push_back(getLightPdf(dist, d.pointOnLight, p));
                put3(oNrm, calcLightNormal(rect, p));
                put3(oNrm, calcLightNormal(sph, p));
            }
        dump("lights_in.f32", in);
        dump("lights_rect_uniform.f32", oU);
        dump("lights_rect_sph.f32", oS);
        dump("lights_sphere.f32", oSph);
        dump("lights_distant.f32", oD);
        dump("lights_pdf.f32", oPdf);
        dump("lights_normal.f32", oNrm);
        std::vector<float> mis;
        for (int i = 0; i < 32; ++i)
        {
            const float a = 0.01f + 10.0f * frand(), b = 5.0f * frand();
            mis.insert(mis.end(), { a, b, misWeightBalance(a, b) });
        }
        dump("mis.f32", mis);
        std::vector<float> area = { calcLightArea(rect), calcLightArea(sph), calcLightArea(dist) };
        dump("lights_area.f32", area);
    }
    // ---- 3. accumulation: invTM(lerp(TM(prev), TM(new), 1/(i+1))) composed as OptixRender.cu:60-78 does ----
    {
        const float3 e = make_float3(6.25e-4f);
        std::vector<float> in, out, tm;
        float3 prev = make_float3(0.0f);
        for (uint32_t i = 0; i < 64; ++i)
        {
            const float3 v = make_float3(40.0f * frand() * frand(), 10.0f * frand(), 300.0f * frand() * frand() * frand());
            float3 acc = v;
            if (i > 0)
            {
                const float a = 1.0f / static_cast<float>(i + 1);
                acc = inverseTonemap(lerp(tonemap(prev, e), tonemap(acc, e), a), e);
            }
            prev = acc;
            put3(in, v);
            put3(out, acc);
            put3(tm, tonemap(v, e));
            put3(tm, inverseTonemap(tonemap(v, e), e));
        }
        dump("accum_in.f32", in);
        dump("accum_out.f32", out);
        dump("tonemap.f32", tm);
        // the SURVEY probe: e = 0.0625
        const float3 e2 = make_float3(0.0625f);
        const float3 r =
            inverseTonemap(lerp(tonemap(make_float3(1, 2, 3), e2), tonemap(make_float3(3, 2, 1), e2), 0.5f), e2);
        std::vector<float> probe;
        put3(probe, r);
        dump("accum_probe.f32", probe);
    }
    // ---- 4. sutil helpers the restatement depends on ----
    {
        std::vector<float> v;
        for (int i = 0; i < 16; ++i)
        {
            const float3 a = make_float3(frand() * 4 - 2, frand() * 4 - 2, frand() * 4 - 2);
            const float s = 0.1f + 3.0f * frand();
            put3(v, a);
            v.push_back(s);
            put3(v, a / s);
            put3(v, normalize(a));
            v.push_back(length(a));
            v.push_back(clamp(a.x, -1.0f, 1.0f));
        }
        dump("sutil.f32", v);
    }
    return 0;
}
