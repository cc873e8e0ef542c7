// SNAPSHOT (round 4, commit 7fd1c86; not built): k_shade<HAIR, LATE> -- the build that shades continuation (TailQ) rays from their records.
// ------------------------------------------------------------------------------------------------------------
// k_shade: __miss__ms (OptixRender.cu:250-257), __closesthit__light (:315-341), __closesthit__radiance
// (closest_hit.cu:456-606) and the tail of the raygen bounce loop (OptixRender.cu:131-153) for one bounce.
// ------------------------------------------------------------------------------------------------------------
#ifndef SKH_SHADE_ATTR
#define SKH_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(4, 4))) // 128 VGPRs (4 spilled dwords): four 256-thread blocks per CU
#endif
#ifndef SKH_SHADE_BLOCK
#define SKH_SHADE_BLOCK 256 // 132 VGPRs = 3 waves/SIMD: 256-thread blocks (1 wave per SIMD) fill all three, 512-thread blocks only two
#endif
// HAIR: the build with df::chiang_hair_bsdf in it (launched when the material list holds a hair material); LATE: the build that knows about
// continuations (TailQ: parked queue entries are skipped, a ray's bounce index is launchDepth - its lag, the last workgroups shade late rays
// from their records) -- passes without continuations run the build without it (its id words are plain path ids)
template <bool HAIR, bool LATE = false>
__global__ void __launch_bounds__(SKH_SHADE_BLOCK) SKH_SHADE_ATTR
    k_shade(DevScene sc, FrameP fp, uint32_t sampleOffset, uint32_t launchDepth /* bounce index of a ray = launchDepth - its lag */, const uint32_t* __restrict__ tileXY, RayQ rq,
            const uint32_t* __restrict__ countPtr, HitQ hq, PathS ps, RayQ nextQ, uint32_t* __restrict__ nextCount, RayQ shadowQ,
            float* __restrict__ contrib, uint32_t* __restrict__ shadowCount,
            uint32_t lateBlocks /* the LAST lateBlocks workgroups shade the "late" rays: parked by the launch before, resumed by this bounce's closest-hit launch ... */,
            const uint32_t* __restrict__ lateRec /* ... read from their records (TailQ) ... */, const uint32_t* __restrict__ lateCount /* ... a list per shard ... */,
            uint32_t lateCap /* ... of this capacity */)
{
    __shared__ uint32_t s_wave[2 * (SKH_COMPACT_MAX_WAVES + 1)];
    __shared__ uint32_t s_sobol[SKH_SOBOL_LUT_WORDS];
#if SKH_MATERIALS_LDS
    // north_star: "material params staged through LDS": the first SKH_MATERIALS_LDS argument blocks (64 B each) ride along with the
    // Sobol table; a hit whose material lies beyond them reads global memory as before
    __shared__ float4 s_mat[SKH_MATERIALS_LDS * 4];
#endif
    // workgroup b works on shard b & 7 (and compacts into the same shard of both output queues); the late workgroups split the parked rays'
    // per-shard lists the same way: a ray is shaded into the output shard of its input shard
    const bool lateMode = LATE && blockIdx.x >= gridDim.x - lateBlocks;
    const uint32_t bIdx = lateMode ? blockIdx.x - (gridDim.x - lateBlocks) : blockIdx.x;
    const uint32_t shard = bIdx & (SKH_SHARDS - 1u), lb = bIdx / SKH_SHARDS;
    const uint32_t n = lateMode ? min(lateCount[shard * SKH_COUNT_STRIDE], lateCap) : countPtr[shard * SKH_COUNT_STRIDE]; // rays in this shard
    if (lb * blockDim.x >= n)
        return; // whole block past the end of its shard
    const uint32_t il = lb * blockDim.x + threadIdx.x;
    const uint32_t i = lateMode ? shard * lateCap + il : shard * rq.region + il;
#ifdef SKH_LANE_PROFILE
    unsigned long long spc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, spT = __builtin_readcyclecounter();
#define SKH_SP(k)                                                    \
    {                                                                \
        const unsigned long long t_ = __builtin_readcyclecounter();  \
        spc[k] += t_ - spT;                                          \
        spT = t_;                                                    \
    }
#else
#define SKH_SP(k)
#endif
    {
        // 20 KB table -> LDS: the block's five fetches go out together (a rolled loop waited for each in turn)
        constexpr int passes = ((SKH_SOBOL_LUT_WORDS / 4) + SKH_SHADE_BLOCK - 1) / SKH_SHADE_BLOCK; // (256 threads: five whole passes; 512: the third is half one)
        uint4 lut[passes];
#pragma unroll
        for (int k = 0; k < passes; ++k)
            if ((k + 1) * SKH_SHADE_BLOCK <= SKH_SOBOL_LUT_WORDS / 4 || threadIdx.x + k * SKH_SHADE_BLOCK < SKH_SOBOL_LUT_WORDS / 4)
                lut[k] = reinterpret_cast<const uint4*>(g_sobol_lut)[threadIdx.x + k * SKH_SHADE_BLOCK];
            else
                lut[k] = make_uint4(0u, 0u, 0u, 0u);
#if SKH_MATERIALS_LDS
        float4 mrow = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        static_assert(SKH_MATERIALS_LDS * 4 <= SKH_SHADE_BLOCK, "one float4 of the material table per thread");
        if (threadIdx.x < SKH_MATERIALS_LDS * 4 && threadIdx.x < sc.numMaterials * 4u)
            mrow = reinterpret_cast<const float4*>(sc.materials)[threadIdx.x];
#endif
#pragma unroll
        for (int k = 0; k < passes; ++k)
            if ((k + 1) * SKH_SHADE_BLOCK <= SKH_SOBOL_LUT_WORDS / 4 || threadIdx.x + k * SKH_SHADE_BLOCK < SKH_SOBOL_LUT_WORDS / 4)
                reinterpret_cast<uint4*>(s_sobol)[threadIdx.x + k * SKH_SHADE_BLOCK] = lut[k];
#if SKH_MATERIALS_LDS
        if (threadIdx.x < SKH_MATERIALS_LDS * 4)
            s_mat[threadIdx.x] = mrow;
#endif
    }
    __syncthreads();
    bool valid = il < n;
    bool emitNext = false, emitShadow = false;
    v3 nextO = mk3(0.0f), nextD = mk3(0.0f), shO = mk3(0.0f), shD = mk3(0.0f), shC = mk3(0.0f);
    float shTmax = 0.0f;
    uint32_t pid = 0, lag = 0;
    uint32_t lateWord = 0;
    if (valid)
    {
        // the id word: path | lag << 28 | parked << 31.  A ray the closest-hit launch parked has no hit yet (the next launch resumes it, the late
        // workgroups of the NEXT k_shade shade it); a late record that was parked again has moved on to the next list
        const uint32_t idw = lateMode ? TailQ::plane(const_cast<uint32_t*>(lateRec), lateCap, 0)[i] : rq.ids()[i];
        if (LATE)
        {
            if (lateMode)
                lateWord = TailQ::plane(const_cast<uint32_t*>(lateRec), lateCap, 2)[i];
            if (lateMode ? lateWord == 0xffffffffu : (idw & SKH_PARKED_BIT) != 0u)
                valid = false;
            pid = idw & SKH_PATH_MASK;
            lag = (idw >> SKH_LAG_SHIFT) & 7u;
        }
        else
            pid = idw;
    }
    const uint32_t depth = launchDepth - lag; // (lag <= launchDepth: a ray is parked at most once per launch)
    if (valid)
    {
        v3 rayO, rayD;
        float4 hr0, hr1;
        if (lateMode)
        {
            uint32_t* R = const_cast<uint32_t*>(lateRec);
            rayO = mk3(__uint_as_float(TailQ::plane(R, lateCap, 9)[i]), __uint_as_float(TailQ::plane(R, lateCap, 10)[i]), __uint_as_float(TailQ::plane(R, lateCap, 11)[i]));
            rayD = mk3(__uint_as_float(TailQ::plane(R, lateCap, 12)[i]), __uint_as_float(TailQ::plane(R, lateCap, 13)[i]), __uint_as_float(TailQ::plane(R, lateCap, 14)[i]));
            hr0 = make_float4(__uint_as_float(TailQ::plane(R, lateCap, 3)[i]), __uint_as_float(TailQ::plane(R, lateCap, 4)[i]), __uint_as_float(TailQ::plane(R, lateCap, 5)[i]), 0.0f);
            hr1 = make_float4(__uint_as_float(TailQ::plane(R, lateCap, 6)[i]), __uint_as_float(TailQ::plane(R, lateCap, 7)[i]), 0.0f, 0.0f);
        }
        else
        {
            rayO = mk3(rq.plane(0)[i], rq.plane(1)[i], rq.plane(2)[i]);
            rayD = mk3(rq.plane(3)[i], rq.plane(4)[i], rq.plane(5)[i]);
            hr0 = hq.rec(i)[0], hr1 = hq.rec(i)[1];
        }
        const float ht = hr0.x, hu = hr0.y, hv = hr0.z;
        const uint32_t hinst = __float_as_uint(hr1.x), hprim = __float_as_uint(hr1.y);
        float* P = ps.base;
        const size_t S = ps.stride;
        // (depth 0: the PerRayData initial values, OptixRender.cu:96-109 -- k_raygen does not store them)
        v3 throughput = depth == 0u ? mk3(1.0f) : mk3(P[pid], P[pid + S], P[pid + 2 * S]);
        // prd.radiance stays in the path state and is read-modify-written only by the branches that change it (a light hit, the debug and error
        // colours; the miss program's `+= throughput * 0` only when that product is not zero, i.e. a non-finite throughput): most paths of most
        // bounces leave it alone, and 12 B read + 12 B written per path were a tenth of this kernel's traffic.  Same values in the same order.
        v3 radiance = mk3(0.0f);
        bool radianceDirty = false;
#define SKH_RADIANCE_LOAD() radiance = mk3(P[pid + 3 * S], P[pid + 4 * S], P[pid + 5 * S]), radianceDirty = true
        float lastBsdfPdf = depth == 0u ? 0.0f : P[pid + 6 * S];
        uint32_t flags = depth == 0u ? 0u : reinterpret_cast<uint32_t*>(P)[pid + 7 * S];
        bool inside = (flags & PF_INSIDE) != 0;
        bool specularBounce = (flags & PF_SPECULAR) != 0;
        uint32_t firstEvent = (flags >> PF_EVENT_SHIFT) & 3u;
        uint32_t px, py;
        const uint32_t sub = pid / fp.numSlots;
        slot_to_pixel(fp, tileXY, pid - sub * fp.numSlots, px, py);
        Sampler smp = init_sampler(px, py, fp.subframeIndex + sampleOffset + sub, fp.sppTotal, 52u);
        smp.depth = depth; // prd.sampler.depth++ once per bounce (OptixRender.cu:153)
        uint32_t prdDepth = depth;
        v3 origin = rayO, dir = rayD; // prd.origin / prd.dir keep their old value when no hit program sets them

        if (hinst == 0xffffffffu)
        {
            // __miss__ms: bg_color = 0 (OptixRender.cpp:739)
            const v3 bg = throughput * mk3(0.0f);
            if (!(bg.x == 0.0f && bg.y == 0.0f && bg.z == 0.0f))
            {
                SKH_RADIANCE_LOAD();
                radiance = radiance + bg;
            }
            throughput = mk3(0.0f);
            prdDepth = fp.maxDepth;
        }
        else
        {
            SKH_SP(0) // queue / path-state loads, sampler
            const HostInstance hi = sc.instances[hinst];
            const float* w2o = sc.inst[hinst].w2o;
            // A hit on a baked triangle names its shading record itself (SKH_PRIM_DIRECT, k_gather_tris): the 96-byte fetch -- the one that
            // misses the caches -- goes out BESIDE the instance record's instead of behind it (chain: queue -> {instance, triangle} -> material,
            // was queue -> instance -> {triangle, material}).  Other hits read record 0 here for nothing and theirs below.
            const bool directTv = (hprim & SKH_PRIM_DIRECT) != 0u;
            float4 tv[6];
            {
                const float4* tp = sc.shadeTris + 6 * (size_t)(directTv ? (hprim & ~SKH_PRIM_DIRECT) : 0u);
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    tv[k] = tp[k];
            }
            // (the whole record now: the compiler sinks the loads of `material` / `light` below the type test = one more round trip)
            asm volatile("" ::"v"(hi.type), "v"(hi.material), "v"(hi.light));
            if (hi.type == 1)
            {
                // __closesthit__light
                const Light& l = sc.lights[hi.light < sc.numLights ? hi.light : 0u]; // (skh_build_accel validates it; the light list may have been replaced since)
                const v3 hitPoint = rayO + ht * rayD;
                const v3 lightNormal = calc_light_normal(l, hitPoint);
                if (-dot(rayD, lightNormal) > 0.0f)
                {
                    SKH_RADIANCE_LOAD();
                    if (depth == 0 || specularBounce)
                        radiance = radiance + throughput * mk3(l.color) * -dot(rayD, lightNormal);
                    else
                    {
                        const float lightPdf = get_light_pdf(l, hitPoint, rayO) / (float)sc.numLights;
                        const float misWeight = mis_weight_balance(lastBsdfPdf, lightPdf);
                        radiance = radiance + throughput * mk3(l.color) * -dot(rayD, lightNormal) * misWeight;
                    }
                }
                throughput = mk3(0.0f);
            }
            else
            {
                // __closesthit__radiance
                const uint32_t mid = hi.material == 0xffffffffu ? 0u : hi.material; // OptixRender.cpp:768
#if SKH_MATERIALS_LDS
                const uint32_t midc = mid < sc.numMaterials ? mid : 0u;
                Material mat;
                if (midc < (uint32_t)SKH_MATERIALS_LDS)
                {
                    const float4 m0 = s_mat[4 * midc], m1 = s_mat[4 * midc + 1], m2 = s_mat[4 * midc + 2], m3 = s_mat[4 * midc + 3];
                    mat.type = __float_as_uint(m0.x), mat.base_color[0] = m0.y, mat.base_color[1] = m0.z, mat.base_color[2] = m0.w;
                    mat.roughness = m1.x, mat.metallic = m1.y, mat.specular = m1.z, mat.ior = m1.w;
                    mat.base_color_texture = __float_as_uint(m2.x), mat.normal_texture = __float_as_uint(m2.y);
                    mat.reserved[0] = m2.z, mat.reserved[1] = m2.w, mat.reserved[2] = m3.x, mat.reserved[3] = m3.y, mat.reserved[4] = m3.z, mat.reserved[5] = m3.w;
                }
                else
                    mat = sc.materials[midc];
#else
                Material mat = sc.materials[mid < sc.numMaterials ? mid : 0u];
#endif
                // the triangle's shading record goes out together with the material's (both hang off the instance record only);
                // a curve hit fetches record 0 for nothing
                if (!directTv && hi.type != 2)
                {
                    const float4* tp = sc.shadeTris + 6 * (size_t)(hi.light + hprim);
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        tv[k] = tp[k];
                }
                asm volatile("" ::"v"(mat.type), "v"(tv[0].x), "v"(tv[2].x), "v"(tv[4].x));
                // mdlcode_init (closest_hit.cu:507): texture lookups of the material, triangle hits only.  OmniPBR: a valid
                // diffuse_texture replaces the constant colour; a valid normalmap_texture replaces state.normal by
                // normalize(tu x + tv y + n z), (x, y, z) = 2 rgb - 1 (base::tangent_space_normal_texture, factor 1)
                const bool useBase = mat.base_color_texture != 0u && mat.base_color_texture <= sc.numTextures;
                const bool useNormal = mat.normal_texture != 0u && mat.normal_texture <= sc.numTextures;
                const bool textured = hi.type != 2 && (useBase || useNormal);
                SurfaceTex st;
                v3 stT = mk3(0.0f);
                // (a hair material on a triangle mesh reads state.tangent_u too: the vertex tangent, closest_hit.cu:399-400)
                const bool hairOnMesh = HAIR && mat.type == 3u && hi.type != 2;
                SurfaceHit sh = hi.type == 2 ? fill_curve(sc, hi, w2o, hprim, hu, ht, rayO, rayD, inside, HAIR ? &stT : nullptr) :
                                               fill_triangle(hi, w2o, tv, hu, hv, inside, (textured || hairOnMesh) ? &st : nullptr);
                if (hairOnMesh)
                    stT = st.tangent_u;
                if (textured)
                {
                    if (useBase)
                    {
                        const v4 c = tex_lookup_rgba8(sc.texels, sc.texDesc[mat.base_color_texture - 1u], st.u, st.v);
                        mat.base_color[0] = c.x, mat.base_color[1] = c.y, mat.base_color[2] = c.z;
                    }
                    if (useNormal)
                    {
                        const v4 c = tex_lookup_rgba8(sc.texels, sc.texDesc[mat.normal_texture - 1u], st.u, st.v);
                        const v3 ts = mk3(c.x * 2.0f - 1.0f, c.y * 2.0f - 1.0f, c.z * 2.0f - 1.0f);
                        sh.normal = normalize((st.tangent_u * ts.x + st.tangent_v * ts.y) + sh.normal * ts.z);
                    }
                }
                if (fp.debug == 1)
                    radiance = (sh.normal + mk3(1.0f)) * 0.5f, radianceDirty = true;
                else
                {
                    const float xi0 = sampler_random_lut(smp, DIM_BSDF0, s_sobol), xi1 = sampler_random_lut(smp, DIM_BSDF1, s_sobol),
                                xi2 = sampler_random_lut(smp, DIM_BSDF2, s_sobol);
                    const float xi3 = HAIR ? sampler_random_lut(smp, DIM_BSDF3, s_sobol) : 0.0f; // (only the hair BSDF consumes xi.w)
                    const v3 k1 = -rayD;
                    BsdfSample bs;
                    SKH_SP(1) // hit reconstruction, material, textures, bsdf randoms
                    bsdf_sample<HAIR>(mat, sh.normal, sh.geom_normal, stT, k1, xi0, xi1, xi2, xi3, inside, bs);
                    SKH_SP(2) // bsdf_sample
                    if (bs.event_type == EV_ABSORB)
                    {
                        if (depth == 0)
                            firstEvent = 1; // eAbsorb
                        throughput = mk3(0.0f);
                    }
                    else
                    {
                        specularBounce = (bs.event_type & EV_SPECULAR) != 0;
                        if (depth == 0)
                        {
                            if (bs.event_type & EV_DIFFUSE)
                                firstEvent = 2;
                            if (bs.event_type & EV_GLOSSY)
                                firstEvent = 3;
                        }
                        bool errorOut = false;
                        if (bs.event_type & (EV_DIFFUSE | EV_GLOSSY))
                        {
                            // estimateDirectLighting + sampleLight: closest_hit.cu:260-324
                            v3 toLight = mk3(0.0f);
                            float lightPdf = 0.0f;
                            v3 lrad = mk3(0.0f);
                            bool wantShadow = false;
                            float distToLight = 0.0f;
                            if (sc.numLights > 0)
                            {
                                const float u = sampler_random_lut(smp, DIM_LIGHT_ID, s_sobol);
                                const uint32_t lightId = (uint32_t)((float)sc.numLights * u);
                                const float lightSelectionPdf = 1.0f / (float)sc.numLights;
                                // the whole 112-byte record in one round trip (by reference its fields were fetched in three dependent
                                // steps: type, then the branch's points, then colour / normal)
                                const Light light = sc.lights[lightId];
                                asm volatile("" ::"v"(light.points[0].x), "v"(light.points[1].x), "v"(light.points[2].x), "v"(light.points[3].x),
                                             "v"(light.color.x), "v"(light.normal.x), "v"(light.type));
                                const float ux = sampler_random_lut(smp, DIM_LIGHT_X, s_sobol), uy = sampler_random_lut(smp, DIM_LIGHT_Y, s_sobol);
                                LightSample d;
                                d.pointOnLight = mk3(0.0f);
                                d.pdf = 0.0f;
                                d.normal = mk3(0.0f);
                                d.area = 0.0f;
                                d.L = mk3(0.0f);
                                d.distToLight = 0.0f;
                                switch (light.type)
                                {
                                case 0:
                                    d = fp.rectMethod == 0 ? sample_rect_light_uniform(light, ux, uy, sh.position) :
                                                             sample_rect_light(light, ux, uy, sh.position);
                                    break;
                                case 2:
                                    d = sample_sphere_light(light, ux, uy, sh.position);
                                    break;
                                case 3:
                                    d = sample_distant_light(light, ux, uy);
                                    break;
                                default:
                                    break;
                                }
                                toLight = d.L;
                                const v3 Li = mk3(light.color);
                                if (dot(sh.normal, d.L) > 0.0f && -dot(d.L, d.normal) > 0.0f && all3(Li))
                                {
                                    wantShadow = true;
                                    distToLight = d.distToLight;
                                    lightPdf = d.pdf;
                                    lrad = 1.0f * Li * saturatef(dot(sh.normal, d.L)); // visibility applied by k_trace<shadow>
                                }
                                lightPdf *= lightSelectionPdf;
                            }
                            if (isnan3(lrad) || isnan(lightPdf))
                            {
                                radiance = mk3(10000.0f, 0.0f, 0.0f), radianceDirty = true;
                                throughput = mk3(0.0f);
                                errorOut = true;
                            }
                            else
                            {
                                const bool isNextEventValid = ((dot(toLight, sh.normal) > 0.0f) != inside) && lightPdf != 0.0f;
                                if (isNextEventValid)
                                {
                                    BsdfEval ev;
                                    SKH_SP(3) // light sampling
                                    bsdf_evaluate<HAIR>(mat, sh.normal, sh.geom_normal, stT, k1, toLight, inside, ev);
                                    SKH_SP(4) // bsdf_evaluate
                                    if (isnan3(ev.bsdf_diffuse) || isnan3(ev.bsdf_glossy))
                                    {
                                        radiance = mk3(10000.0f, 0.0f, 0.0f), radianceDirty = true;
                                        throughput = mk3(0.0f);
                                        errorOut = true;
                                    }
                                    else if (ev.pdf > 0.0f && wantShadow)
                                    {
                                        const v3 radianceOverPdf = lrad / lightPdf;
                                        const float misWeight = mis_weight_balance(lightPdf, ev.pdf);
                                        shC = throughput * radianceOverPdf * misWeight * (ev.bsdf_diffuse + ev.bsdf_glossy);
                                        shO = offset_ray(sh.position, sh.geom_normal);
                                        shD = toLight;
                                        shTmax = distToLight;
                                        emitShadow = true;
                                    }
                                }
                            }
                        }
                        if (!errorOut)
                        {
                            if (bs.event_type & EV_TRANSMISSION)
                            {
                                inside = !inside;
                                origin = offset_ray(sh.position, -sh.geom_normal);
                            }
                            else
                                origin = offset_ray(sh.position, sh.geom_normal);
                            lastBsdfPdf = specularBounce ? 1.0f : bs.pdf;
                            dir = bs.k2;
                            throughput = throughput * bs.bsdf_over_pdf;
                        }
                    }
                }
            }
        }
        SKH_SP(5) // rest of the hit program
        // tail of the bounce loop: OptixRender.cu:131-153
        bool alive = true;
        if (prdDepth > 3)
        {
            const float p = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z));
            if (sampler_random_lut(smp, DIM_RR, s_sobol) > p)
                alive = false;
            else
                throughput = throughput * (1.0f / (p + 1e-5f));
        }
        if (alive && dot(throughput, throughput) < 1e-5f)
            alive = false;
        if (alive)
        {
            ++prdDepth;
            if (fp.debug == 1)
                alive = false;
        }
        if (alive && prdDepth >= fp.maxDepth)
            alive = false;
        emitNext = alive;
        nextO = origin;
        nextD = dir;
        // write back path state
        P[pid] = throughput.x;
        P[pid + S] = throughput.y;
        P[pid + 2 * S] = throughput.z;
        if (radianceDirty)
        {
            P[pid + 3 * S] = radiance.x;
            P[pid + 4 * S] = radiance.y;
            P[pid + 5 * S] = radiance.z;
        }
#undef SKH_RADIANCE_LOAD
        P[pid + 6 * S] = lastBsdfPdf;
        reinterpret_cast<uint32_t*>(P)[pid + 7 * S] =
            (inside ? PF_INSIDE : 0u) | (specularBounce ? PF_SPECULAR : 0u) | (firstEvent << PF_EVENT_SHIFT);
    }
    SKH_SP(6) // bounce tail + path-state write
    // stream compaction of live paths / shadow rays: one atomic per queue per workgroup, both in flight together
    uint32_t ni, si;
    block_compact2(emitNext, nextCount + shard * SKH_COUNT_STRIDE, emitShadow, shadowCount + shard * SKH_COUNT_STRIDE, s_wave, ni, si);
    ni += shard * nextQ.region; // (a shard's output never outgrows its region: at most one ray of either kind per input ray)
    si += shard * shadowQ.region;
    if (emitNext)
    {
        nextQ.plane(0)[ni] = nextO.x;
        nextQ.plane(1)[ni] = nextO.y;
        nextQ.plane(2)[ni] = nextO.z;
        nextQ.plane(3)[ni] = nextD.x;
        nextQ.plane(4)[ni] = nextD.y;
        nextQ.plane(5)[ni] = nextD.z;
        // (planes 6 / 7, tmin / tmax: constants of the pass, filled once by the host -- k_fill_f32 in render_one)
        nextQ.ids()[ni] = LATE ? (pid | (lag << SKH_LAG_SHIFT)) : pid; // (what a late ray emits stays one launch behind its bounce index)
    }
    if (emitShadow)
    {
        shadowQ.plane(0)[si] = shO.x;
        shadowQ.plane(1)[si] = shO.y;
        shadowQ.plane(2)[si] = shO.z;
        shadowQ.plane(3)[si] = shD.x;
        shadowQ.plane(4)[si] = shD.y;
        shadowQ.plane(5)[si] = shD.z;
        // (plane 6 = shadowTmin: filled once by the host)
        shadowQ.plane(7)[si] = shTmax;
        shadowQ.ids()[si] = pid;
        contrib[si] = shC.x;
        contrib[si + shadowQ.stride] = shC.y;
        contrib[si + 2 * (size_t)shadowQ.stride] = shC.z;
    }
#ifdef SKH_LANE_PROFILE
    SKH_SP(7) // compaction + queue writes
    for (int k = 0; k < 8; ++k)
    {
        const uint32_t hi32 = wave_max((uint32_t)(spc[k] >> 4));
        if ((threadIdx.x & 63u) == 0)
            atomicAdd(&sc.profile->shade[k], (unsigned long long)hi32 << 4);
    }
#endif
#undef SKH_SP
}
