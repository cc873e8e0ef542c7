// strelka_hip -- the wavefront kernels (gfx950).
//
// One sub-frame of the reference's megakernel (__raygen__rg + hit programs, src/render/optix/OptixRender.cu:80-248,
// OptixRender_radiance_closest_hit.cu:456-606) is split into per-bounce kernels over SoA queues in HBM:
//
//   k_raygen          camera rays for every owned pixel slot           -> ray queue[0]
//   per bounce b:
//     k_trace<closest> BVH traversal (TLAS -> instance -> BLAS)         ray queue[b] -> hit records
//     k_shade          miss / light hit / surface hit: hit reconstruction, BSDF sample, next-event estimation,
//                      Russian roulette, wave-ballot compaction          -> ray queue[b+1], shadow queue[b]
//     k_trace<shadow>  any-hit traversal; un-occluded contributions are added to the path's radiance
//   k_collect / k_finalize   per-pixel sample sums, AOVs, LDR-space accumulation (OptixRender.cu:60-78,169-247)
//
// Queue layout: SoA, one float/uint plane per field, so that a wave's 64 lanes read 256 contiguous bytes per
// field.  Traversal stack: per-lane, 20 entries in LDS laid out [entry][lane] (conflict-free), spilling to a
// per-thread global overflow area only for pathological depths.
#pragma once
#include "skh_bvh.h"

namespace skh
{

struct DevInstance // 64 B: what traversal needs when it enters an instance
{
    float w2o[12]; // rows of R^-1 (world -> object, 3x3) with the OBJECT-TO-WORLD translation in the fourth column: xform_point_rel
    int rootRef;
    uint32_t mask; // GEOMETRY_MASK_* (OptixRenderParams.h:9-17); 0 = disabled (degenerate transform / empty geometry)
    uint32_t type; // 0 mesh, 1 light, 2 curve
    uint32_t pad;
};
struct HostInstance // == skh_instance (64 B), as uploaded
{
    float o2w[12];
    uint32_t type, geom, material, light;
};

// Curve trees the world-only curve kernel walks one after the other (table entries; the MERGED tree of all identity-transform instances is one).  Round 5 allowed
// 16; measured in round 6 against the two-level kernel on the hair stand-in cut into N prims under translations (gpurun_out/r6j): N = 2 -4 %, 4 -11 %, 8 -20 % --
// every ray visits every tree, nothing culls an instance the ray misses.  Two entries keep "the merged groom + one moved prim" and "one prim under a transform".
#define SKH_WORLD_CURVES 2
#define SKH_SEG_STRIDE 8 // float4 per curve leaf record (DevScene::segs)
#define SKH_REF_CURVEROOT 0x40000000 // stack entry (world-only kernel with curves): (ref & 0xffff) indexes DevScene::worldCurveRoot / worldCurveInst
#define SKH_REF_CURVEROOT_IDENT 0x20000000 // ... of an instance under a bit-exact identity transform, which is only valid as the FIRST curve tree of a ray (o, d still the world ray)
struct DevScene
{
    const Node4* tlasNodes;
    const uint32_t* tlasInst; // leaf order -> instance id
    int tlasRoot;
    uint32_t numWorldCurves; // curve instances under identity transforms, walked from the world-only kernel (no TLAS leaf): their curve trees' roots ...
    int worldCurveRoot[SKH_WORLD_CURVES];
    uint32_t worldCurveInst[SKH_WORLD_CURVES]; // ... and instance ids
    uint32_t worldCurveIdentLast; // 1: the LAST entry's instance sits under a bit-exact identity transform (the host puts such an instance last)
    uint32_t worldCurveMerged; // bit k: entry k is a MERGED group of curve instances under one transform -- worldCurveInst[k] lends the transform, a hit takes its instance from the segment's leaf record
    uint32_t numInstances;
    const DevInstance* inst; // per instance (shading side: w2o)
    const DevInstance* tinst; // per TLAS leaf: (instance, BLAS subtree) after opening; pad = instance id
    const Node4* triNodes;
    const float4* tris; // 3 x float4 per triangle, leaf order
    int worldRoot; // root (inside triNodes) of the group of BAKED mesh instances: world-space triangles {v0, prim | v1, instance | v2, 0} that every
                   // ray walks first, in world space, with no instance entry; SKH_REF_INVALID = nothing baked
    int lightRoot; // the same for baked light proxies: radiance rays only (shadow rays do not see lights)
    const Node4* segNodes;
    // ONE 128-byte record per curve sub-segment, leaf order (round 6: control points, bounding cylinder and ids were three arrays -- a leaf visit fetched the cylinder's
    // line, a candidate then the control points' line and, accepted, the ids': the any-hit launches of the hair stand-in miss the L2 on 24.5 lines per ray, 0.69 of the
    // random-line rate).  float4 [0..3] the segment's four control points {xyz, radius} (duplicated per sub-range: one record per test), [4] [5] the conservative bounding
    // cylinder of the (padded) sub-range {A, R} {unit axis, 0}, [6] {segment index inside its curve set | sub-range << 28, instance of a MERGED segment or ~0, -, -}, [7] -
    const float4* segs;
    uint32_t curveSplit; // parameter sub-ranges per segment (sub-range in word [6].x >> 28 of the leaf record)
    // shading side
    const HostInstance* instances; // shading copy: for mesh instances `light` holds the mesh's first record in shadeTris
    const float4* shadeTris; // de-indexed shading records, 96 B per triangle (k_gather_shade_tris), meshes back to back
    const uint8_t* verts;
    const uint32_t* indices;
    const uint4* meshes;
    const uint32_t* curveSegBase; // per curve set: first entry in segStartAll
    const uint32_t* segStartAll; // per segment: index of its first control point
    const float* cpoints;
    const float* cradii;
    const Light* lights;
    uint32_t numLights;
    const Material* materials;
    uint32_t numMaterials;
    const HairConst* hairConst; // per material: what of df::chiang_hair_bsdf depends on the material only (k_hair_consts), read by the hair build of k_shade
    const uint32_t* texels; // RGBA8 texels of all textures
    const uint4* texDesc; // per texture: {offset in texels, width, height, 0}
    uint32_t numTextures;
    struct StatsDev* profile; // (lane-profile build only)
    uint32_t* overflowFlag; // host-mapped word: set when a traversal stack had to drop an entry (checked after every render / trace call)
};

// Queues are SHARDED 8 ways (SKH_SHARDS = the XCD count): shard g owns the positions [g * region, g * region + count[g]) of every plane.
// A k_shade workgroup b reads and compacts into shard b & 7 -- one queue-tail word per shard and queue, each in its own 128-byte
// line -- and the k_trace waves with blockIdx & 7 == g pull from shard g first: workgroups are dealt round-robin to the 8 XCDs, so a
// ray is written, traced and shaded under the same L2.  One tail word for the whole queue took 256 K returning atomics per launch at
// the ~88 per microsecond one line sustains: 2.9 ms of a 4.0 ms k_shade launch (round 3).
#define SKH_SHARDS 8u
struct RayQ // SoA planes of `stride` elements: ox oy oz dx dy dz tmin tmax pathId  (36 B / ray)
{
    float* base;
    uint32_t stride;
    uint32_t region; // positions per shard (stride = SKH_SHARDS * region)
    __device__ float* plane(int k) const
    {
        return base + (size_t)k * stride;
    }
    __device__ uint32_t* ids() const
    {
        return reinterpret_cast<uint32_t*>(base + (size_t)8 * stride);
    }
};
// Hit records: ONE 32-byte record per ray {t, u, v, 0 | instance, primitive, 0, 0}.  Results are written by whichever lanes finished
// since the last refill -- scattered queue positions --, so five 4-byte planes left five partly written lines per ray (73 B of
// write traffic per 20-byte hit, round-2 counters); a record is one aligned 32-byte sector, and k_shade reads it as two dwordx4.
// (An any-hit launch in raw query mode uses `base` as one float plane: 1 = occluded, -1 = not.)
struct HitQ
{
    float* base;
    uint32_t stride; // rays the buffer holds
    // primBits = B > 0 (render passes of scenes the world-only kernels trace -- every mesh hit there names its shading record -- whose instance count and primitive range share 32 bits: option compact_hits):
    // ONE 16-byte record per ray {t, u, v, instance << B | shading record (a light proxy's primitive index, a curve segment's)}, ~0 in the last word = a miss --
    // 16 B less written per ray by the closest-hit launches and read back by k_shade
    uint32_t primBits;
    uint32_t direct;   // (primBits > 0) 1: the word of a mesh hit is its shading record's index (SKH_PRIM_DIRECT implied); 0: the mesh-local primitive index
    uint32_t recClamp; // (primBits > 0) the last shading record: k_shade's early fetch must stay inside the table when the word is a curve segment's index
    __device__ float4* rec(uint32_t i) const
    {
        return reinterpret_cast<float4*>(base) + 2 * (size_t)i;
    }
    __device__ float4* rec16(uint32_t i) const
    {
        return reinterpret_cast<float4*>(base) + (size_t)i;
    }
};
struct PathS // per path slot: planes 0-2 throughput rgb, 6 lastBsdfPdf, 7 flags (3-5 unused); behind them prd.radiance as ONE float4 per path
{
    float* base;
    uint32_t stride;
    // A path's radiance is read-modify-written from scattered lanes (the any-hit launches add an unoccluded light sample's contribution, a light hit its
    // emission): one 16-byte access per path instead of three 4-byte ones in three planes (a scattered access costs by the instruction: docs/LOG.md, round 5).
    __device__ float4* rad() const
    {
        return reinterpret_cast<float4*>(base + 8 * (size_t)stride);
    }
};
#define SKH_PATH_FLOATS 12 // floats per path the buffer holds: 8 planes + the float4
enum
{
    PF_INSIDE = 1,
    PF_SPECULAR = 2,
    PF_EVENT_SHIFT = 2 // 2 bits: EventType (OptixRenderParams.h:70-77)
};

struct FrameP // skh_frame_params + launch geometry
{
    float viewToWorld[16];
    float clipToView[16];
    uint32_t subframeIndex, samplesThisLaunch, sppTotal, maxDepth, rectMethod;
    float exposure[3];
    uint32_t enableAccumulation, debug;
    float shadowTmin, materialTmin;
    uint32_t width, height, tileSize, tileShift, numTiles, numSlots;
    uint32_t batch; // sub-frames in flight in this wavefront (>= 1): path p = sub * numSlots + slot
    uint32_t finalFirst, finalCount; // k_finalize_batch applies the accumulation steps of sub-frames [finalFirst, finalFirst + finalCount) of the pass
};

#ifndef SKH_STACK_LDS
#define SKH_STACK_LDS 20 // per-lane stack entries kept in LDS: 5 KB per wave, 28 waves per CU fit 160 KB (tests build a variant with 12 to exercise the overflow path)
#endif
#ifndef SKH_STACK_OVF
#define SKH_STACK_OVF 104
#endif
#define SKH_TAIL_EXTRA 8 // slots every overflow column has beyond SKH_STACK_OVF: where the SPLIT builds' tail phase keeps the stack entries whose LDS its family tables take
#define SKH_TRACE_BLOCK 64

// Reciprocal ray direction for the SLAB tests only: v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE division.  Box
// tests need to be conservative, not exact -- hit records come from the primitive tests, which never see `inv` -- and the
// acceptance slack below covers the extra ulp: per plane the computed t carries a relative error <= 2^-23 (rcp) + 3 * 2^-24
// (difference, product, fma), entry and exit of different axes can err in opposite directions, so tnear <= tfar * (1 + 2^-19)
// keeps every box the exact arithmetic would accept (plus the 2^-20 relative inflation of the stored boxes).
// A direction component that is exactly (or nearly) zero must not become an infinite reciprocal: the node test evaluates plane
// distances as q * (cell * inv) + (origin - o) * inv, and inf - inf = NaN there makes min/max drop the whole axis -- the box is
// accepted whatever the ray's position on that axis.  Still conservative, but one axis-parallel ray (a mirror bounce off a
// vertical face: d.y == 0) then walks thousands of nodes on its own and the persistent launch waits for it: one such ray in
// 64 M cost a launch 7.7 ms (found with the lane-profile build).  Clamped to +-2^-60 the products stay finite (cell sizes and
// scene extents are far below 2^60) and the slab test decides the parallel axis by the ray's position, as it should.
SKH_DI float rcp_safe(float x)
{
    return __builtin_amdgcn_rcpf(copysignf(fmaxf(fabsf(x), 0x1p-60f), x)); // v_max_f32 |x|, v_bfi_b32, v_rcp_f32
}
SKH_DI v3 rcp3(const v3& d)
{
    return mk3(rcp_safe(d.x), rcp_safe(d.y), rcp_safe(d.z));
}
#define SKH_SLAB_SLACK 1.0000019073486328125f // 1 + 2^-19

struct TraceCounters
{
    uint32_t nodes, prims, segs, insts;
};

SKH_DI bool slab_test(const v3& lo, const v3& hi, const v3& o, const v3& inv, float tmin, float tmax, float& tnear)
{
    const float t0x = (lo.x - o.x) * inv.x, t1x = (hi.x - o.x) * inv.x;
    const float t0y = (lo.y - o.y) * inv.y, t1y = (hi.y - o.y) * inv.y;
    const float t0z = (lo.z - o.z) * inv.z, t1z = (hi.z - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    tnear = tn;
    return tn <= tf * 1.0000002384185791015625f;
}

struct HitRec
{
    float t;
    uint32_t inst, prim;
    float u, v;
    bool found;
};

struct StatsDev
{
    unsigned long long raysRadiance, raysShadow, nodes[2], prims[2], segs[2], insts[2];
#ifdef SKH_LANE_PROFILE
    // wave-level event counts of k_trace (profile build only): [0] node-loop iterations, [1] triangle-loop iterations,
    // [2] instance-entry blocks, [3] outer iterations, [4] refills, [5] lanes refilled, [6] leaf blocks, [7] pop blocks
    unsigned long long wave[2][10]; // ... [6] triangle passes with an fp64 fallback, [7] with a division, [8] with a passed sign test
    unsigned long long shade[8]; // k_shade cycle split (SKH_SP marks)
    unsigned int slowCount, slowPad; // rays that took more than SKH_SLOW_RAY node steps: the first 16 are recorded
    float slow[16][12]; // steps, tris, insts, kernel, o xyz, d xyz, tmin, tmax
    unsigned long long runHist[2][42], blockHist[2][42]; // Newton steps per run / of the longest run of a block (curve builds)
    unsigned long long cyc[2][10]; // summed over waves: [0] refill [1] node loop [2] leaf [3] pop [4] result write [5] whole kernel [6] cycles [7] 100 MHz ticks [8] the curve block
#endif
#ifdef SKH_TAIL_PROFILE
    // launch tails (a build of its own: a few atomics per WAVE, nothing in the loops), in 16-microsecond bins from the launch's first wave (launchT0: reset by the
    // host before every trace launch; 100 MHz ticks): when the waves found the queue dry, when they left, how long after their dry point, how many lanes they had
    // alive then, and the ray-ticks they spent after it (a ray alive for one tick)
    unsigned long long launchT0[2], dryHist[2][64], exitHist[2][64], afterDryHist[2][64], dryLive[2][65], rayTicksAfterDry[2], waveTicksAfterDry[2];
#endif
};

// number of set bits of `mask` in the lanes below this one: v_mbcnt_lo / v_mbcnt_hi on the ballot's two scalar halves -- two instructions and
// no per-lane "lanes below me" mask to keep (popcount(mask & ((1 << lane) - 1)) holds that 64-bit mask in two VGPRs or rebuilds it with a 64-bit shift)
SKH_DI uint32_t rank_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
SKH_DI uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v = max(v, (uint32_t)__shfl_xor(v, off));
    return v;
}

SKH_DI uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off);
    return v;
}

#ifndef SKH_CURVE_MIN_WAVES
#define SKH_CURVE_MIN_WAVES 6 // 80 VGPRs (29 dwords spilled) for the build with the inlined curve intersector: hair stand-in 425 / 455 / 478 / 480 Mray/s
                              // at 4 / 5 / 6 / 7 waves per SIMD (the build needs 104 VGPRs unconstrained)
#endif
#ifndef SKH_WORLD_CURVE_ANYHIT_MIN_WAVES
#define SKH_WORLD_CURVE_ANYHIT_MIN_WAVES 6
#endif
#ifndef SKH_WORLD_CURVE_MIN_WAVES
#define SKH_WORLD_CURVE_MIN_WAVES 6 // the world-only kernel with the curve block
#endif
#ifndef SKH_ANYHIT_MIN_WAVES
#define SKH_ANYHIT_MIN_WAVES 7 // the any-hit build needs 71 VGPRs: 28 waves per CU (shadow 45.9 -> 43.4 ms over 24)
#endif
#ifndef SKH_TRACE_MIN_WAVES
#define SKH_TRACE_MIN_WAVES 7 // 72 VGPRs (2 dwords spilled): the launch runs 28 one-wave blocks per CU (closest 103.3 -> 100.6 ms over 6 waves at 77 VGPRs)
#endif
#ifndef SKH_MATERIALS_LDS
#define SKH_MATERIALS_LDS 64 // material argument blocks k_shade stages in LDS beside the Sobol table (0 = all from global memory: measured equal, 31.3 vs 31.2 ms -- the fetch was never on the critical path; on because north_star asks for it)
#endif
#ifndef SKH_WORLD_ANYHIT_MIN_WAVES
#define SKH_WORLD_ANYHIT_MIN_WAVES 8 // the world-only any-hit build fits 64 VGPRs without scratch (59; 27 SGPRs go to lanes): 32 waves per CU -- any-hit 28.7 -> 27.85 ms
                                     // without mesh sharing, 22.85 -> 21.95 architectural, 34.5 -> 34.4 on the kitchen (docs/LOG.md round 4)
#endif
#ifndef SKH_WORLD_CLOSEST_MIN_WAVES
#define SKH_WORLD_CLOSEST_MIN_WAVES 8 // the world-only closest-hit build at 64 VGPRs (one dword of the refill path in scratch, 16 SGPRs in lanes) and 32 waves per CU:
                                      // closest-hit 71.3 -> 70.2 ms without mesh sharing, 56.55 -> 55.35 architectural, 81.9 -> 81.95 kitchen.  It took lane ranks by
                                      // v_mbcnt (rank_below) and the overflow area addressed where it is used to get there: with the 64-bit "lanes below me" mask
                                      // and the overflow pointer spilled (5 dwords, two reloads per outer iteration) the same build ran 81.7 -> 85.7 ms
#endif
#ifndef SKH_TRACE_ATTR
#define SKH_TRACE_ATTR
#endif
#ifndef SKH_TRI_COOP
#define SKH_TRI_COOP 1 // 1: the triangle pass of the world-only builds is shared -- lanes that are NOT at a leaf take the second triangle of the
                       // two-triangle leaves (the owner's ray pulled with ds_bpermute, their own ray state parked in the free part of their LDS stack
                       // column meanwhile), so that one pass does what took two at 30 + 17 of 64 lanes
#endif
// The 8 ray-fetch cursors of a launch sit in separate 128-byte lines: returning atomics on ONE line serialise at ~88 per
// microsecond chip-wide (measured), which eight cursors in the same line would share.
#define SKH_FETCH_STRIDE 32
#define SKH_COUNT_STRIDE 32 // same for the queue-length words the compaction atomics hit

// ------------------------------------------------------------------------------------------------------------
// k_trace: persistent waves over the ray queue, two-level BVH traversal (TLAS -> instance -> BLAS).
//
// Work distribution: the queue is cut into 8 contiguous ranges, one per XCD label (blockIdx % 8; blocks b and b+8
// share an XCD under the observed round-robin placement, so neighbouring rays -- similar BVH working set -- land in
// one XCD's L2).  A wave pulls rays from its range through one returning atomic per refill and steals from the next
// ranges when its own is empty.  Lanes whose ray has terminated are refilled as soon as `fetchMin` lanes are idle
// ("persistent while-while with dynamic fetch"), which keeps the 64-wide wave populated when ray lengths diverge.
// Placement and fetch order only affect speed: every ray's result is independent of scheduling.
//
// Closest hit = smallest t, ties broken by the smaller (instance, primitive) key, ray interval open at both ends:
// the result does not depend on the BVH or on the traversal order (DESIGN.md "determinism").
// ------------------------------------------------------------------------------------------------------------
// WORLD: the build for scenes whose every instance is baked (no TLAS leaf, no curve set -- what a bake without mesh sharing gives,
// HdStrelka's per-instance meshes): one world-space tree, no instance entry / exit, no object-space copy of the ray, no sentinel.
// (The measured-negative variants of round 4 -- pop-time culling, postponed leaves, the touch prefetch, packed node FMAs, 8-wide nodes, continuations --
// live in experiments/skh_trace_r04_variants.h with their numbers; three builds ship: world-only, two-level, two-level + curves.)
// (keeps a loop-invariant address out of the hoister's hands: the overflow column's 64-bit address is built where it is used -- branches that
// never run unless a stack passes its LDS entries -- instead of living in two registers, or in scratch, through the traversal loops)
SKH_DI uint32_t skh_opaque(uint32_t v)
{
    asm volatile("" : "+s"(v)); // (a wave-uniform value: stays scalar)
    return v;
}
struct LightBox
{
    float lo[3], hi[3];
};
#ifndef SKH_BEST_LDS
#define SKH_BEST_LDS 1
#endif
#ifndef SKH_SEGNODE
#define SKH_SEGNODE 0 // 1: the curve builds understand SEGMENT NODES (skh_bvh.h k_segnode_emit; option curve_segnode).  A measured negative of round 6 (hair
                      // 2 003 -> 1 835 Mray/s: +15 node visits per shadow ray for no fewer Newton runs, docs/LOG.md): compiled out of the default
                      // library, built and held to the oracle's hit records by tests/test_gpu_parity.py through a -DSKH_SEGNODE=1 variant
#endif
#ifndef SKH_ANYHIT_FLAT_PUSH
#define SKH_ANYHIT_FLAT_PUSH 1 // the any-hit node loop pushes its hit children without a branch per child (A/B: -DSKH_ANYHIT_FLAT_PUSH=0)
#endif
template <bool ANY_HIT, bool COUNT, bool CURVES, bool WORLD = false, bool SPLIT = false>
__global__ void __launch_bounds__(SKH_TRACE_BLOCK, CURVES ? (WORLD ? (ANY_HIT ? SKH_WORLD_CURVE_ANYHIT_MIN_WAVES : SKH_WORLD_CURVE_MIN_WAVES) : SKH_CURVE_MIN_WAVES) : (WORLD ? (ANY_HIT ? SKH_WORLD_ANYHIT_MIN_WAVES : SKH_WORLD_CLOSEST_MIN_WAVES) : (ANY_HIT ? SKH_ANYHIT_MIN_WAVES : SKH_TRACE_MIN_WAVES))) SKH_TRACE_ATTR
    k_trace(DevScene sc, RayQ rq, const uint32_t* __restrict__ countPtr, uint32_t* __restrict__ fetch /*8 counters, zeroed*/,
            uint32_t fetchArg /* refill threshold | curve-test threshold << 8 | node-break threshold << 16 | leaf-kind threshold << 24 */, 
            HitQ hq, PathS ps, const float4* __restrict__ contrib /* per shadow-queue position: {contribution rgb, path id} */, int* __restrict__ ovfBase,
            StatsDev* __restrict__ stats, LightBox lightBox /* around the baked light proxies (closest-hit builds) */, uint32_t fetchChunk /* world-only triangle builds: queue positions reserved per atomic, 0 = what each refill needs */)
{
    // WORLD && CURVES (round 5): the world-only kernel with the curve block in it -- scenes whose every mesh instance is baked and that hold at
    // most SKH_WORLD_CURVES curve instances.  Their curve trees' roots wait at the BOTTOM of every ray's stack as markers
    // (SKH_REF_CURVEROOT | k): a ray walks the world-space triangles first, then each curve tree, with no top level, no instance entry block, no
    // sentinel and no world-space copy of the ray in registers (hair stand-in: an instance-entry pass in 98 % of the outer iterations before).
    // Taking a marker sends the world ray (re-read from the queue) through the instance's transform exactly as a TLAS leaf would: hit records are
    // the two-level path's, bit for bit.
    constexpr bool TRICOOP = SKH_TRI_COOP && WORLD && !CURVES; // (closest-hit and any-hit builds of the world-only TRIANGLE kernel)
    // per-lane stack entries in LDS (the rest: SKH_STACK_OVF entries in global memory); TRICOOP gives one up for its two 64-byte lane tables
    // (LDS is handed out in 1280-byte granules here: 20 x 256 B = 4 granules exactly, 128 B more would cost a fifth = 25 instead of 28 waves per CU)
    // The closest-hit CURVE builds keep the attributes of the best hit so far -- instance, primitive, u, v: written when a hit is accepted, read on a tie
    // and once per ray for the result -- in LDS instead of four registers (they are the builds that spill, and their scratch traffic goes to HBM:
    // docs/LOG.md, round 5); one stack entry pays for the 1 KB (19 x 256 + 512 (s_runs) + 1024 = 5 granules, as before).
    constexpr bool BESTLDS = CURVES && !ANY_HIT && SKH_BEST_LDS;
    // SPLIT (round 6, the world-only triangle builds, small passes): once the queue is dry the idle lanes of a wave take stack entries of the lanes that still
    // have a ray -- a long ray's subtrees are walked side by side instead of one after the other: docs/LOG.md "launch tails".  The fragments of one ray (a
    // FAMILY, named by the lane that held the ray when the wave went dry) merge their results in LDS by the rule of the sequential walk -- nearer, or equally near
    // with the smaller (instance, primitive) key: the record does not depend on who found what -- and the last one to finish writes it.  Seven 256-byte
    // tables (pruning bound, t, instance, primitive, u, v, live fragments) come out of the stack's LDS entries.
    static_assert(!SPLIT || (WORLD && !CURVES), "SPLIT: the world-only triangle builds");
    constexpr int NLDS0 = (TRICOOP || BESTLDS) ? SKH_STACK_LDS - 1 : SKH_STACK_LDS;
    constexpr int NLDST = (SPLIT && NLDS0 >= 11) ? NLDS0 - 7 : NLDS0; // LDS stack entries of the tail phase (the tests' tiny-stack variants keep what little they have: their tables get LDS of their own)
    static_assert(NLDS0 - NLDST <= SKH_TAIL_EXTRA, "the overflow columns' extra slots hold the LDS entries the tail phase gives up");
    __shared__ int s_stack[NLDS0 * SKH_TRACE_BLOCK];
    __shared__ uint32_t s_famx[(SPLIT && NLDST == NLDS0) ? 7 * SKH_TRACE_BLOCK : 1];
    uint32_t* const s_fam = NLDST < NLDS0 ? reinterpret_cast<uint32_t*>(s_stack + NLDST * SKH_TRACE_BLOCK) : s_famx;
    __shared__ uint32_t s_best[BESTLDS ? 4 * SKH_TRACE_BLOCK : 1];
    __shared__ unsigned char s_tab[(TRICOOP || SPLIT) ? 128 : 1]; // [0..63] owner lanes by rank, [64..127] helper lanes by rank (SPLIT: [0..63] giving lanes by rank)
    const uint32_t fetchMin = fetchArg & 0xffu, curveMin = (fetchArg >> 8) & 0xffu, nodeBreak = (fetchArg >> 16) & 0xffu, leafMin = fetchArg >> 24;
    const uint32_t lane = threadIdx.x;
    uint32_t n = 0; // (countPtr: SKH_SHARDS queue-length words, SKH_COUNT_STRIDE apart)
#pragma unroll
    for (uint32_t g = 0; g < SKH_SHARDS; ++g)
        n += countPtr[g * SKH_COUNT_STRIDE];
    if (n == 0)
        return;
    const uint32_t perGroup = rq.region;
    const uint32_t group = blockIdx.x & 7u;
    uint32_t tries = 0;
    bool exhausted = false;
    uint32_t resBase = 0, resLeft = 0; // the wave's reservation in the ray queue (fetchChunk): wave-uniform
    uint32_t fam = threadIdx.x; // (SPLIT) the family this lane's ray fragment belongs to
#define SKH_FAM(k, l) s_fam[(k) * SKH_TRACE_BLOCK + (l)] /* 0 bound (bits of a non-negative t) 1 t 2 instance 3 primitive 4 u 5 v 6 fragments alive */
    int* lds = s_stack + lane;
    // (the overflow area is addressed from ovfBase where it is used -- rare paths -- instead of through a per-lane 64-bit pointer held across the loops)
// (stack entry e of a lane: LDS below NLDSP; from NLDS0 on the overflow column's slot e - NLDS0; the tail phase's entries NLDSP .. NLDS0 - 1 -- LDS in the main phase -- SKH_STACK_OVF + e - NLDSP)
#define SKH_OVF_SLOT(e) ((SKH_TAIL_PHASE && (e) < NLDS0) ? SKH_STACK_OVF + ((e) - NLDSP) : (e) - NLDS0)
#define SKH_OVF_AT(e) ovfBase[(size_t)(e) * ovfStride + (skh_opaque(blockIdx.x * SKH_TRACE_BLOCK) + threadIdx.x)]
    const uint32_t ovfStride = gridDim.x * SKH_TRACE_BLOCK;
    const uint32_t rayMask = CURVES ? (ANY_HIT ? 3u : 255u) : (ANY_HIT ? 1u : 253u);
    TraceCounters tc = { 0, 0, 0, 0 };
#ifdef SKH_LANE_PROFILE
    uint32_t wv[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    uint32_t rayNodes = 0, rayTris = 0, rayInsts = 0;
    unsigned long long cy[7] = { 0, 0, 0, 0, 0, 0, 0 };
    const unsigned long long cyStart = __builtin_readcyclecounter();
    const unsigned long long rtStart = __builtin_amdgcn_s_memrealtime(); // constant 100 MHz counter: cycles / realtime = the clock this launch really ran at
#define SKH_LP(...) __VA_ARGS__
#else
#define SKH_LP(...)
#endif
#ifdef SKH_TAIL_PROFILE
    if (threadIdx.x == 0)
        atomicMin(&stats->launchT0[ANY_HIT ? 1 : 0], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    bool dryNoted = false;
    unsigned long long dryLast = 0, dryAt = 0, rayTicks = 0;
    uint32_t dryLiveLanes = 0;
#endif

    // per-lane traversal state
    bool hasRay = false, pending = false;
    uint32_t pend = 0; // (curve build) segments of the current leaf that passed the cheap test and wait for the full one
    uint32_t ridx = 0;
    v3 ow = mk3(0.0f), dw = mk3(0.0f), o = mk3(0.0f), d = mk3(0.0f), inv = mk3(0.0f);
    v3 invw = mk3(0.0f); // world-space reciprocal direction: kept by the any-hit build (67 VGPRs), recomputed at every instance exit by the
                         // closest-hit build, which needs the three registers to stay at 72 = 7 waves per SIMD
    float tmin = 0.0f;
    RayShear sh;
    sh.perm = 0;
    sh.Sx = sh.Sy = sh.Sz = 0.0f;
    const Node4* nodes = sc.tlasNodes;
    bool inBlas = false;
    uint32_t curInst = 0, curType = 0;
    int sp = 0, cur = SKH_REF_INVALID;
    HitRec best;
    best.t = 0.0f, best.inst = best.prim = 0xffffffffu, best.u = best.v = 0.0f, best.found = false;

#define SKH_BEST_INST() (BESTLDS ? s_best[lane] : best.inst)
#define SKH_BEST_PRIM() (BESTLDS ? s_best[SKH_TRACE_BLOCK + lane] : best.prim)
#define SKH_BEST_SET(INST, PRIM, U, V)                                                                \
    {                                                                                                 \
        if (BESTLDS)                                                                                  \
        {                                                                                             \
            s_best[lane] = (INST), s_best[SKH_TRACE_BLOCK + lane] = (PRIM);                           \
            s_best[2 * SKH_TRACE_BLOCK + lane] = __float_as_uint(U), s_best[3 * SKH_TRACE_BLOCK + lane] = __float_as_uint(V); \
        }                                                                                             \
        else                                                                                          \
            best.inst = (INST), best.prim = (PRIM), best.u = (U), best.v = (V);                       \
    }
// a baked light proxy's triangle in the world-space group (merge_light_proxies): any-hit queries do not see lights (the proxies that stay instances
// are masked at their TLAS leaf); k_gather_tris marks a baked proxy's triangles in the last word of the record (0 everywhere else)
#define SKH_HIDDEN_LIGHT(C) (ANY_HIT && __float_as_uint((C).w) != 0u)
#define SKH_PUSH(v)                                                  \
    {                                                                \
        if (sp < NLDSP)                                              \
            lds[sp * SKH_TRACE_BLOCK] = (v);                         \
        else if (sp < NLDS0 + SKH_STACK_OVF)                         \
            SKH_OVF_AT(SKH_OVF_SLOT(sp)) = (v);                      \
        else                                                         \
            *sc.overflowFlag = 1u; /* the entry is dropped: the call that launched this kernel returns SKH_FAIL, never silent */ \
        ++sp;                                                        \
    }
#define SKH_POP(dst)                                                 \
    {                                                                \
        --sp;                                                        \
        if (sp < NLDSP)                                              \
            dst = lds[sp * SKH_TRACE_BLOCK];                         \
        else if (sp < NLDS0 + SKH_STACK_OVF)                         \
            dst = SKH_OVF_AT(SKH_OVF_SLOT(sp));                      \
        else                                                         \
            dst = SKH_REF_INVALID;                                   \
    }

// cheap conservative rejection of a curve leaf's sub-segments: (distance between the ray's line and the segment's bounding cylinder
// axis)^2 = ((A - o) . n)^2 / |n|^2, n = d x axis; a long thin diagonal hair fills a tiny part of its box.  Sets bit k of PEND for sub-segment k that passes.
#define SKH_CYLINDER_TESTS(FIRST, COUNT, PEND)                                                                                   \
    for (uint32_t k_ = 0; k_ < (COUNT); ++k_)                                                                                    \
    {                                                                                                                            \
        const float4 b0 = sc.segs[SKH_SEG_STRIDE * (size_t)((FIRST) + k_) + 4], b1 = sc.segs[SKH_SEG_STRIDE * (size_t)((FIRST) + k_) + 5]; \
        const v3 w = mk3(b0.x - o.x, b0.y - o.y, b0.z - o.z);                                                                    \
        const v3 nn = cross(d, mk3(b1.x, b1.y, b1.z));                                                                           \
        const float n2 = dot(nn, nn), wn = dot(w, nn);                                                                           \
        const float Rm = b0.w + (fabsf(w.x) + fabsf(w.y) + fabsf(w.z)) * 4e-6f; /* cancellation in w . n */                      \
        if (n2 > 1e-12f * dot(d, d) && wn * wn > Rm * Rm * n2 * 1.0001f)                                                         \
            continue;                                                                                                            \
        (PEND) |= 1u << k_;                                                                                                      \
    }
#define SKH_TAKE_MARKER()                                                                                         \
    {                                                                                                             \
        const uint32_t k = (uint32_t)cur & 0xffffu;                                                               \
        const uint32_t xinst = sc.worldCurveInst[k]; /* whose transform the entry is entered through */          \
        curInst = ((sc.worldCurveMerged >> k) & 1u) ? 0xffffffffu : xinst; /* ~0: the segment record names the instance */ \
        if ((uint32_t)cur & SKH_REF_CURVEROOT_IDENT)                                                              \
        {                                                                                                         \
            /* a bit-exact identity transform (the common bake): the world ray is still in (o, d); it goes through the identity as it would */ \
            /* through the instance's record (x * 1 + y * 0 + z * 0 can turn a -0 into +0), without the nine loads below */ \
            const float ident[12] = { 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f };    \
            o = xform_point_rel(ident, o);                                                                        \
            d = xform_vector(ident, d);                                                                           \
        }                                                                                                         \
        else                                                                                                      \
        {                                                                                                         \
            const float4* ip = reinterpret_cast<const float4*>(sc.inst + xinst);                                  \
            const float4 i0 = ip[0], i1 = ip[1], i2 = ip[2];                                                      \
            const v3 wo = mk3(rq.plane(0)[ridx], rq.plane(1)[ridx], rq.plane(2)[ridx]);                           \
            const v3 wd = mk3(rq.plane(3)[ridx], rq.plane(4)[ridx], rq.plane(5)[ridx]);                           \
            const float m[12] = { i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w };       \
            o = xform_point_rel(m, wo);                                                                           \
            d = xform_vector(m, wd);                                                                              \
        }                                                                                                         \
        inv = rcp3(d);                                                                                            \
        curType = 2;                                                                                              \
        cur = sc.worldCurveRoot[k];                                                                               \
    }

    // The main phase: every build.  (SPLIT: left as soon as the wave finds the queue dry.)
    {
        constexpr int NLDSP = NLDS0;
        constexpr bool TRICOOPP = TRICOOP;
#define SKH_TAIL_PHASE 0
        for (;;)
        {
#include "skh_trace_body.inc"
        }
#undef SKH_TAIL_PHASE
    }
    if constexpr (SPLIT)
    {
        // The tail phase (SPLIT builds): the same pass without the refill; the family tables take the place of the upper LDS stack entries, which move to
        // SKH_TAIL_EXTRA more slots at the end of the lane's overflow column (a lane rarely has that many).
        constexpr int NLDSP = NLDST;
#ifndef SKH_TAIL_TRICOOP
#define SKH_TAIL_TRICOOP 1
#endif
        constexpr bool TRICOOPP = TRICOOP && SKH_TAIL_TRICOOP;
        if (NLDST < NLDS0)
        {
            for (int e = NLDST; e < NLDS0; ++e)
                if (hasRay && e < sp)
                    SKH_OVF_AT(SKH_STACK_OVF + e - NLDST) = lds[e * SKH_TRACE_BLOCK];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // every ray the wave holds becomes a family of one; its record starts as the lane's best hit so far
        if (hasRay)
        {
            fam = lane;
            SKH_FAM(0, lane) = __float_as_uint(best.t);
            SKH_FAM(1, lane) = __float_as_uint(best.t);
            SKH_FAM(2, lane) = best.found ? (ANY_HIT ? 0u : best.inst) : 0xffffffffu; // (all ones: nothing found yet)
            if constexpr (!ANY_HIT) // (an occlusion query's record is one word: found or not)
                SKH_FAM(3, lane) = best.prim, SKH_FAM(4, lane) = __float_as_uint(best.u), SKH_FAM(5, lane) = __float_as_uint(best.v);
            SKH_FAM(6, lane) = 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#define SKH_TAIL_PHASE 1
        for (;;)
        {
#include "skh_trace_body.inc"
        }
#undef SKH_TAIL_PHASE
    }
#undef SKH_PUSH
#undef SKH_HIDDEN_LIGHT
#undef SKH_POP
#undef SKH_OVF_AT
#undef SKH_OVF_SLOT
#undef SKH_TAKE_MARKER
#undef SKH_CYLINDER_TESTS
#undef SKH_FAM
#ifdef SKH_TAIL_PROFILE
    if (lane == 0 && dryNoted)
    {
        const int A = ANY_HIT ? 1 : 0;
        const unsigned long long t0 = *(volatile unsigned long long*)&stats->launchT0[A], now = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&stats->dryHist[A][min(63ull, (dryAt - t0) / 1600ull)], 1ull);
        atomicAdd(&stats->exitHist[A][min(63ull, (now - t0) / 1600ull)], 1ull);
        atomicAdd(&stats->afterDryHist[A][min(63ull, (now - dryAt) / 1600ull)], 1ull);
        atomicAdd(&stats->dryLive[A][dryLiveLanes], 1ull);
        atomicAdd(&stats->rayTicksAfterDry[A], rayTicks);
        atomicAdd(&stats->waveTicksAfterDry[A], now - dryAt);
    }
#endif
    if (COUNT)
    {
        const uint32_t a = wave_sum(tc.nodes), b = wave_sum(tc.prims), c2 = wave_sum(tc.segs), d2 = wave_sum(tc.insts);
#ifdef SKH_LANE_PROFILE
        cy[5] = __builtin_readcyclecounter() - cyStart;
        if (lane == 0)
        {
            atomicAdd(&stats->cyc[ANY_HIT ? 1 : 0][6], cy[5]);
            atomicAdd(&stats->cyc[ANY_HIT ? 1 : 0][7], __builtin_amdgcn_s_memrealtime() - rtStart);
        }
        for (int k = 0; k < 7; ++k)
        {
            // cycle sums are wave-uniform increments taken by the lanes that were active: the busiest lane has (nearly) all of them
            uint32_t hi = wave_max((uint32_t)(cy[k] >> 8));
            if (lane == 0)
                atomicAdd(&stats->cyc[ANY_HIT ? 1 : 0][k == 6 ? 8 : k], (unsigned long long)hi << 8);
        }
#endif
        if (lane == 0)
        {
            atomicAdd(&stats->nodes[ANY_HIT ? 1 : 0], (unsigned long long)a);
            atomicAdd(&stats->prims[ANY_HIT ? 1 : 0], (unsigned long long)b);
            atomicAdd(&stats->segs[ANY_HIT ? 1 : 0], (unsigned long long)c2);
            atomicAdd(&stats->insts[ANY_HIT ? 1 : 0], (unsigned long long)d2);
#ifdef SKH_LANE_PROFILE
            for (int k = 0; k < 10; ++k)
                atomicAdd(&stats->wave[ANY_HIT ? 1 : 0][k], (unsigned long long)wv[k]);
#endif
        }
    }
#undef SKH_LP
}

// what of the hair BSDF depends on the material only, one thread per material (skh_device.h hair_const; skh_set_materials)
__global__ void k_hair_consts(const Material* __restrict__ mats, uint32_t n, HairConst* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = hair_const(mats[i]);
}

// ------------------------------------------------------------------------------------------------------------
// slot <-> pixel: slot = tile * T^2 + morton(xl, yl): a wave's 64 lanes cover an 8x8 pixel block
// ------------------------------------------------------------------------------------------------------------
SKH_DI bool slot_to_pixel(const FrameP& fp, const uint32_t* __restrict__ tileXY, uint32_t slot, uint32_t& px, uint32_t& py)
{
    const uint32_t tile = slot >> (2 * fp.tileShift);
    const uint32_t m = slot & ((1u << (2 * fp.tileShift)) - 1u);
    px = tileXY[2 * tile] + compact1by1(m);
    py = tileXY[2 * tile + 1] + compact1by1(m >> 1);
    return px < fp.width && py < fp.height;
}

// Stream compaction: wave ballot + prefix popcount inside the wave, one LDS slot per wave, ONE global atomic per
// workgroup (a single queue-tail word sustains only ~88 returning atomics per microsecond on MI355X, so per-wave
// atomics would serialise a 2 M-path launch for ~0.4 ms).  All threads of the block must call it.
#define SKH_COMPACT_MAX_WAVES 8
SKH_DI uint32_t block_compact(bool emit, uint32_t* counter, uint32_t* s_wave /*[SKH_COMPACT_MAX_WAVES + 1]*/)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long m = __ballot(emit);
    if (lane == 0)
        s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < nw; ++w)
        {
            const uint32_t cnt = s_wave[w];
            s_wave[w] = tot;
            tot += cnt;
        }
        s_wave[SKH_COMPACT_MAX_WAVES] = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    const uint32_t r = s_wave[SKH_COMPACT_MAX_WAVES] + s_wave[wave] + rank_below(m);
    __syncthreads(); // s_wave is reused by the next call
    return r;
}
// Two compactions with one round of barriers and both queue-tail atomics in flight together (k_shade emits a continuation
// ray and a shadow ray per path).  s_wave2: [2][SKH_COMPACT_MAX_WAVES + 1].
SKH_DI void block_compact2(bool emitA, uint32_t* counterA, bool emitB, uint32_t* counterB, uint32_t* s_wave2, uint32_t& ia, uint32_t& ib)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long ma = __ballot(emitA), mb = __ballot(emitB);
    uint32_t* sa = s_wave2;
    uint32_t* sb = s_wave2 + SKH_COMPACT_MAX_WAVES + 1;
    if (lane == 0)
    {
        sa[wave] = (uint32_t)__popcll(ma);
        sb[wave] = (uint32_t)__popcll(mb);
    }
    __syncthreads();
    if (threadIdx.x == 0 || threadIdx.x == 64)
    {
        uint32_t* sw = threadIdx.x == 0 ? sa : sb; // wave 0 serves queue A, wave 1 queue B (a one-wave block: thread 0 does both)
        uint32_t tot = 0;
        for (uint32_t w = 0; w < nw; ++w)
        {
            const uint32_t cnt = sw[w];
            sw[w] = tot;
            tot += cnt;
        }
        sw[SKH_COMPACT_MAX_WAVES] = tot ? atomicAdd(threadIdx.x == 0 ? counterA : counterB, tot) : 0u;
    }
    if (nw == 1 && threadIdx.x == 0)
    {
        uint32_t tot = sb[0];
        sb[0] = 0;
        sb[SKH_COMPACT_MAX_WAVES] = tot ? atomicAdd(counterB, tot) : 0u;
    }
    __syncthreads();
    ia = sa[SKH_COMPACT_MAX_WAVES] + sa[wave] + rank_below(ma);
    ib = sb[SKH_COMPACT_MAX_WAVES] + sb[wave] + rank_below(mb);
}

// Ray generation without atomics: which slots of a tile set fall inside the image is known on the host, so the first queue
// position of every 512-slot block (`blockBase`, exclusive prefix of the per-block valid counts) and the number of valid
// pixels per sub-frame (`validPerSub`) are tables; inside a block the rank comes from ballots.  The queue order is the slot
// order (tile-major, Morton inside a tile), sub-frame after sub-frame -- and one returning atomic per block less (a single
// queue-tail word takes ~88 of those per microsecond: a 64 M-path launch was bound by exactly that).
__global__ void __launch_bounds__(512) k_raygen(FrameP fp, const uint32_t* __restrict__ tileXY, uint32_t sampleOffset, RayQ rq,
                                               uint32_t* __restrict__ counter, PathS ps, const uint32_t* __restrict__ blockBase,
                                               uint32_t blocksPerSub, uint32_t validPerSub)
{
    __shared__ uint32_t s_wave[SKH_COMPACT_MAX_WAVES + 1];
    // several sub-frames can be in flight at once (fp.batch): path = sub * numSlots + slot.  Paths of different
    // sub-frames are independent; only the accumulation (k_finalize_batch) has to respect their order.
    const uint32_t sub = blockIdx.x / blocksPerSub, bi = blockIdx.x - sub * blocksPerSub;
    const uint32_t slot = bi * blockDim.x + threadIdx.x;
    const uint32_t path = sub * fp.numSlots + slot;
    uint32_t px = 0, py = 0;
    const bool active = sub < fp.batch && slot < fp.numSlots && slot_to_pixel(fp, tileXY, slot, px, py);
    v3 o = mk3(0.0f), d = mk3(0.0f);
    if (active)
    {
        const Sampler s = init_sampler(px, py, fp.subframeIndex + sampleOffset + sub, fp.sppTotal, 52u); // OptixRender.cu:101
        generate_camera_ray(px, py, fp.width, fp.height, fp.clipToView, fp.viewToWorld, sampler_random(s, DIM_PIXEL_X),
                            sampler_random(s, DIM_PIXEL_Y), o, d);
        // PerRayData init: OptixRender.cu:96-109.  Throughput = 1 and lastBsdfPdf = 0 are not stored: k_shade ASSUMES the initial
        // values at depth 0 instead of reading them, and writes its own for every path of that bounce, misses included
        ps.rad()[path] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        reinterpret_cast<uint32_t*>(ps.base)[path + 7 * (size_t)ps.stride] = 0u; // flags: outside, eUndef (k_collect reads them even when max_depth = 0 launches no k_shade)
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(active);
    if (lane == 0)
        s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w)
        before += s_wave[w];
    // dense index in slot order -> shard: the queue is cut into SKH_SHARDS equal runs of `per` rays (the last one shorter)
    const uint32_t dense = sub * validPerSub + blockBase[bi] + before + rank_below(m);
    const uint32_t total = validPerSub * fp.batch;
    const uint32_t per = (((total + SKH_SHARDS - 1u) / SKH_SHARDS) + 63u) & ~63u;
    const uint32_t shard = per ? dense / per : 0u;
    const uint32_t idx = shard * rq.region + (dense - shard * per);
    if (blockIdx.x == 0 && threadIdx.x < SKH_SHARDS) // the queue lengths the next kernels read
        counter[threadIdx.x * SKH_COUNT_STRIDE] = min(per, total - min(total, threadIdx.x * per));
    if (active)
    {
        rq.plane(0)[idx] = o.x;
        rq.plane(1)[idx] = o.y;
        rq.plane(2)[idx] = o.z;
        rq.plane(3)[idx] = d.x;
        rq.plane(4)[idx] = d.y;
        rq.plane(5)[idx] = d.z;
        // (planes 6 / 7 -- tmin = materialTmin, tmax = 1e16, OptixRender.cu:121-122 -- hold the same two constants for every radiance ray of every
        // bounce: the host fills them once per queue and per value, k_fill_f32 in render_one; nobody rewrites them per ray)
        rq.ids()[idx] = path;
    }
}

// De-indexed shading vertices: the 3 x 32 B of every triangle, meshes back to back (meshTriBase = exclusive prefix of the
// meshes' triangle counts).  165 MB for the 1.7 M triangles of the kitchen stand-in, of 288 GB.
__global__ void __launch_bounds__(256) k_gather_shade_tris(const uint8_t* __restrict__ verts, const uint32_t* __restrict__ indices,
                                                          const uint4* __restrict__ meshes, const uint32_t* __restrict__ meshTriBase,
                                                          uint32_t nMeshes, uint32_t nTris, float4* __restrict__ out)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nTris)
        return;
    uint32_t lo = 0, hi = nMeshes; // meshTriBase[lo] <= g < meshTriBase[hi]
    while (hi - lo > 1u)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if (meshTriBase[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    const uint4 me = meshes[lo];
    const uint32_t t = g - meshTriBase[lo];
    // record: {p0, normal0} {p1, normal1} {p2, normal2} | {tangent0, tangent1, tangent2, uv0} {uv1, uv2, 0, 0} | unused: what every hit needs sits in the first
    // three float4 -- three scattered loads per hit instead of six --, what only textured (or hair-on-mesh) hits need in the next two
    float4 v0[3], v1[3];
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k)
    {
        const uint32_t idx = indices[me.x + 3u * t + k];
        const float4* v = reinterpret_cast<const float4*>(verts + (size_t)(me.z + idx) * 32);
        v0[k] = v[0], v1[k] = v[1];
    }
    float4* o = out + 6 * (size_t)g;
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k)
        o[k] = make_float4(v0[k].x, v0[k].y, v0[k].z, v1[k].x);
    o[3] = make_float4(v0[0].w, v0[1].w, v0[2].w, v1[0].y);
    o[4] = make_float4(v1[1].y, v1[2].y, 0.0f, 0.0f);
    o[5] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

struct SurfaceHit
{
    v3 position, normal, geom_normal;
};
struct SurfaceTex // state.text_coords[0], tangent_u[0], tangent_v[0]: only textured materials consume them
{
    float u, v;
    v3 tangent_u, tangent_v;
};

// fillTriangleGeomData: closest_hit.cu:365-421.  UVs / tangent frame are computed only when `tex` is given.
// The reference walks mesh -> 3 indices -> 3 vertices (closest_hit.cu:365-376); here the three vertices of every triangle sit
// de-indexed in one 96-byte record (k_gather_shade_tris) whose mesh base came with the instance record -- one dependent fetch
// instead of three -- and k_shade issues that fetch together with the material's (`tv` = the record).
SKH_DI SurfaceHit fill_triangle(const HostInstance& hi, const float* w2o, const float4* tv /* {p, normal} x 3 */, const float4* tx /* {tangent x 3, uv0} {uv1, uv2} */, float bu, float bv, bool inside, SurfaceTex* tex)
{
    const v3 p0 = mk3(tv[0]), p1 = mk3(tv[1]), p2 = mk3(tv[2]);
    const v3 n0 = unpack_normal(__float_as_uint(tv[0].w)), n1 = unpack_normal(__float_as_uint(tv[1].w)),
             n2 = unpack_normal(__float_as_uint(tv[2].w));
    SurfaceHit s;
    s.position = xform_point(hi.o2w, interpolate_attrib(p0, p1, p2, bu, bv));
    const v3 object_normal = interpolate_attrib(n0, n1, n2, bu, bv);
    const v3 worldNormal = normalize(xform_normal(w2o, object_normal));
    v3 geomNormal = cross(p1 - p0, p2 - p0);
    geomNormal = normalize(xform_normal(w2o, geomNormal));
    const float flip = inside ? -1.0f : 1.0f;
    s.geom_normal = geomNormal * flip;
    s.normal = worldNormal * flip;
    if (tex)
    {
        float u0, v0u, u1, v1u, u2, v2u;
        unpack_uv(__float_as_uint(tx[0].w), u0, v0u);
        unpack_uv(__float_as_uint(tx[1].x), u1, v1u);
        unpack_uv(__float_as_uint(tx[1].y), u2, v2u);
        const float bw = 1.0f - bu - bv;
        tex->u = (u0 * bw + u1 * bu) + u2 * bv;
        tex->v = (v0u * bw + v1u * bu) + v2u * bv;
        const v3 t0 = unpack_normal(__float_as_uint(tx[0].x)), t1 = unpack_normal(__float_as_uint(tx[0].y)), t2 = unpack_normal(__float_as_uint(tx[0].z));
        // the tangent goes through the NORMAL transform in the reference (closest_hit.cu:399-400)
        tex->tangent_u = normalize(xform_normal(w2o, interpolate_attrib(t0, t1, t2, bu, bv)));
        tex->tangent_v = cross(s.normal, tex->tangent_u); // worldBinormal, with the already flipped normal (:404)
    }
    return s;
}
// fillCurveGeomData: closest_hit.cu:423-454
SKH_DI SurfaceHit fill_curve(const DevScene& sc, const HostInstance& hi, const float* w2o, uint32_t prim, float u, float t,
                             const v3& rayO, const v3& rayD, bool inside, v3* tangent_u /*state.tangent_u[0], hair build only*/)
{
    const uint32_t s0 = sc.segStartAll[sc.curveSegBase[hi.geom] + prim];
    v4 q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        const float* p = sc.cpoints + 3 * (size_t)(s0 + k);
        q[k] = mk4(p[0], p[1], p[2], sc.cradii[s0 + k]);
    }
    CubicPoly ip;
    cubic_from_bspline(ip, q);
    v3 hitPoint = rayO + t * rayD;
    hitPoint = xform_point_rel(w2o, hitPoint);
    v3 worldNormal = normalize(xform_normal(w2o, curve_surface_normal(ip, u, hitPoint)));
    worldNormal = worldNormal * (inside ? -1.0f : 1.0f);
    if (tangent_u) // curveTangent through the normal transform: closest_hit.cu:436-437, curve.h:412-417
        *tangent_u = normalize(xform_normal(w2o, normalize(mk3(cubic_velocity(ip, u)))));
    SurfaceHit s;
    s.position = xform_point(hi.o2w, hitPoint);
    s.normal = worldNormal;
    s.geom_normal = worldNormal;
    return s;
}

// ------------------------------------------------------------------------------------------------------------
// k_shade: __miss__ms (OptixRender.cu:250-257), __closesthit__light (:315-341), __closesthit__radiance
// (closest_hit.cu:456-606) and the tail of the raygen bounce loop (OptixRender.cu:131-153) for one bounce.
// ------------------------------------------------------------------------------------------------------------
#ifndef SKH_SHADE_WAVES
#define SKH_SHADE_WAVES(HAIR) ((HAIR) ? 4 : 5) // waves per SIMD: the triangle-material build fits 96 VGPRs without a spill since round 5 (the path's radiance stays in memory, the
                                                // continuation code is gone): kitchen k_shade 30.7 -> 27.8 ms, Cornell 8.97 -> 8.20; six waves spill 27 dwords (30.2 ms); the hair build
                                                // (Chiang BSDF) spills 27 at five (16.7 -> 18.2 ms) and stays at four
#endif
#define SKH_SHADE_ATTR(HAIR) __attribute__((amdgpu_waves_per_eu(SKH_SHADE_WAVES(HAIR), SKH_SHADE_WAVES(HAIR))))
#ifndef SKH_SHADE_BLOCK
#define SKH_SHADE_BLOCK 256 // 132 VGPRs = 3 waves/SIMD: 256-thread blocks (1 wave per SIMD) fill all three, 512-thread blocks only two
#endif
// HAIR: the build with df::chiang_hair_bsdf in it (launched when the material list holds a hair material)
template <bool HAIR>
__global__ void __launch_bounds__(SKH_SHADE_BLOCK) SKH_SHADE_ATTR(HAIR)
    k_shade(DevScene sc, FrameP fp, uint32_t sampleOffset, uint32_t depth /* bounce index */, const uint32_t* __restrict__ tileXY, RayQ rq,
            const uint32_t* __restrict__ countPtr, HitQ hq, PathS ps, RayQ nextQ, uint32_t* __restrict__ nextCount, RayQ shadowQ,
            float4* __restrict__ contrib, uint32_t* __restrict__ shadowCount)
{
    __shared__ uint32_t s_wave[2 * (SKH_COMPACT_MAX_WAVES + 1)];
    __shared__ uint32_t s_sobol[SKH_SOBOL_LUT_WORDS];
#if SKH_MATERIALS_LDS
    // north_star: "material params staged through LDS": the first SKH_MATERIALS_LDS argument blocks (64 B each) ride along with the
    // Sobol table; a hit whose material lies beyond them reads global memory as before
    __shared__ float4 s_mat[SKH_MATERIALS_LDS * 4];
#endif
    // workgroup b works on shard b & 7 (and compacts into the same shard of both output queues)
    const uint32_t shard = blockIdx.x & (SKH_SHARDS - 1u), lb = blockIdx.x / SKH_SHARDS;
    const uint32_t n = countPtr[shard * SKH_COUNT_STRIDE]; // rays in this shard
    if (lb * blockDim.x >= n)
        return; // whole block past the end of its shard
    const uint32_t il = lb * blockDim.x + threadIdx.x;
    const uint32_t i = shard * rq.region + il;
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
    bool valid = il < n;
    uint32_t pid = 0;
    v3 rayO = mk3(0.0f), rayD = mk3(0.0f);
    float4 hr0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), hr1 = hr0;
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
    bool emitNext = false, emitShadow = false;
    v3 nextO = mk3(0.0f), nextD = mk3(0.0f), shO = mk3(0.0f), shD = mk3(0.0f), shC = mk3(0.0f);
    float shTmax = 0.0f;
    if (valid)
    {
        pid = rq.ids()[i];
        rayO = mk3(rq.plane(0)[i], rq.plane(1)[i], rq.plane(2)[i]);
        rayD = mk3(rq.plane(3)[i], rq.plane(4)[i], rq.plane(5)[i]);
        if (hq.primBits != 0u)
        {
            // the 16-byte record of a world-only triangle scene: every mesh hit there names its shading record (SKH_PRIM_DIRECT); a light proxy's
            // primitive index is not looked at by its hit program
            hr0 = *hq.rec16(i);
            const uint32_t w = __float_as_uint(hr0.w);
            hr1.x = __uint_as_float(w == 0xffffffffu ? w : w >> hq.primBits);
            hr1.y = __uint_as_float(w == 0xffffffffu ? w : ((w & ((1u << hq.primBits) - 1u)) | (hq.direct ? SKH_PRIM_DIRECT : 0u)));
        }
        else
            hr0 = hq.rec(i)[0], hr1 = hq.rec(i)[1];
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
#define SKH_RADIANCE_LOAD() radiance = mk3(ps.rad()[pid]), radianceDirty = true
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
            float4 tv[3], tx[2];
            // (tangents / UVs: only a textured material or a hair material on a mesh reads them -- scenes without either do not fetch them)
            const bool txWanted = HAIR || sc.numTextures != 0u;
            tx[0] = tx[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            {
                // (a 16-byte record -- HitQ::primBits -- marks every hit as direct: a curve segment's index must not leave the table)
                const uint32_t recIdx = directTv ? (hprim & ~SKH_PRIM_DIRECT) : 0u;
                const float4* tp = sc.shadeTris + 6 * (size_t)(hq.primBits ? min(recIdx, hq.recClamp) : recIdx);
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    tv[k] = tp[k];
                if (txWanted)
                    tx[0] = tp[3], tx[1] = tp[4];
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
                    for (int k = 0; k < 3; ++k)
                        tv[k] = tp[k];
                    if (txWanted)
                        tx[0] = tp[3], tx[1] = tp[4];
                }
                asm volatile("" ::"v"(mat.type), "v"(tv[0].x), "v"(tv[1].x), "v"(tv[2].x));
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
                SurfaceHit sh = hi.type == 2 ? fill_curve(sc, hi, w2o, hprim & ~SKH_PRIM_DIRECT /* (a segment index never has the bit; a 16-byte record sets it for every hit) */, hu, ht, rayO, rayD, inside, HAIR ? &stT : nullptr) :
                                               fill_triangle(hi, w2o, tv, tx, hu, hv, inside, (textured || hairOnMesh) ? &st : nullptr);
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
                    // estimateDirectLighting + sampleLight: closest_hit.cu:260-324 -- as a block of its own: a hit of a HAIR material (hair build) samples the light FIRST, so that
                    // the BSDF's two evaluations for this k1 -- the sampled direction's and the light sample's -- share the fibre geometry (hair_sample_and_evaluate); every
                    // other hit samples it where the reference does, behind the BSDF sample's event type.  The values do not depend on the order.
                    v3 toLight = mk3(0.0f);
                    float lightPdf = 0.0f;
                    v3 lrad = mk3(0.0f);
                    bool wantShadow = false;
                    float distToLight = 0.0f;
                    auto sampleLight = [&]() {
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
                    };
                    const bool hairHit = HAIR && mat.type == 3u;
                    BsdfEval evHair;
                    bool evHairDone = false;
                    if (hairHit)
                    {
                        sampleLight();
                        // (the evaluation is wanted exactly where the flow below would ask for it, should the sampled event not be an absorption)
                        evHairDone = !(isnan3(lrad) || isnan(lightPdf)) && (((dot(toLight, sh.normal) > 0.0f) != inside) && lightPdf != 0.0f);
                        hair_sample_and_evaluate(mat, sh.normal, sh.geom_normal, stT, k1, xi0, xi1, xi2, xi3, sc.hairConst + (mid < sc.numMaterials ? mid : 0u), bs, evHairDone, toLight, evHair);
                    }
                    else
                        bsdf_sample<HAIR>(mat, sh.normal, sh.geom_normal, stT, k1, xi0, xi1, xi2, xi3, inside, bs, HAIR ? sc.hairConst + (mid < sc.numMaterials ? mid : 0u) : nullptr);
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
                            if (!hairHit)
                                sampleLight();
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
                                    if (hairHit)
                                        ev = evHair; // (evaluated beside the sample: evHairDone holds exactly when this branch is reached)
                                    else
                                        bsdf_evaluate<HAIR>(mat, sh.normal, sh.geom_normal, stT, k1, toLight, inside, ev, HAIR ? sc.hairConst + (mid < sc.numMaterials ? mid : 0u) : nullptr);
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
            ps.rad()[pid] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
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
        nextQ.ids()[ni] = pid;
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
        // (the path id travels with the contribution -- one 16-byte record the any-hit launch reads back with ONE scattered load -- not in the queue's id plane)
        contrib[si] = make_float4(shC.x, shC.y, shC.z, __uint_as_float(pid));
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

// a constant plane of a ray queue (tmin / tmax of the radiance queues, tmin of the shadow queue): written when the queues are allocated or the
// value changes, instead of once per ray per bounce by k_raygen / k_shade
__global__ void __launch_bounds__(256) k_fill_f32(float* __restrict__ p, size_t n, float v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = v;
}

// ------------------------------------------------------------------------------------------------------------
// per-pixel sample sums (the `result += prd.radiance` loop of OptixRender.cu:154-167) and the AOV / accumulation
// epilogue (OptixRender.cu:169-247).  sums: result rgb, diffuse rgb, specular rgb, diffuseSamples, specularSamples
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_collect(FrameP fp, const uint32_t* __restrict__ tileXY, uint32_t sampleOffset, PathS ps,
                                                float* __restrict__ sums)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t px, py;
    if (slot >= fp.numSlots || !slot_to_pixel(fp, tileXY, slot, px, py))
        return;
    const size_t S = ps.stride, N = fp.numSlots;
    const v3 rad = mk3(ps.rad()[slot]);
    const uint32_t fe = (reinterpret_cast<const uint32_t*>(ps.base)[slot + 7 * S] >> PF_EVENT_SHIFT) & 3u;
    float v[11];
    if (sampleOffset == 0)
    {
#pragma unroll
        for (int k = 0; k < 11; ++k)
            v[k] = 0.0f;
    }
    else
    {
#pragma unroll
        for (int k = 0; k < 11; ++k)
            v[k] = sums[slot + k * N];
    }
    v[0] += rad.x, v[1] += rad.y, v[2] += rad.z;
    if (fe == 2)
    {
        v[3] += rad.x, v[4] += rad.y, v[5] += rad.z;
        v[9] += 1.0f;
    }
    if (fe == 3)
    {
        v[6] += rad.x, v[7] += rad.y, v[8] += rad.z;
        v[10] += 1.0f;
    }
#pragma unroll
    for (int k = 0; k < 11; ++k)
        sums[slot + k * N] = v[k];
}

// A pixel's accumulation state -- accumulator, the two AOVs and their sample counters -- in registers: k_finalize_batch applies up to 64 accumulation steps to it
// and used to read and write all of it in global memory at every step (1.68 ms of the 64-sub-frame frame for 2.7 GB of radiance reads).  `dirty`: what a step wrote
// (1 accumulator, 2 diffuse AOV + counter, 4 specular AOV + counter): only that goes back.
struct PixState
{
    float4 accum, diffuse, specular;
    uint32_t dcnt, scnt, dirty;
};
SKH_DI PixState load_pix(uint32_t slot, const float4* __restrict__ accum, const float4* __restrict__ diffuse, const float4* __restrict__ specular,
                         const uint16_t* __restrict__ diffuseCounter, const uint16_t* __restrict__ specularCounter)
{
    PixState st;
    st.accum = accum[slot], st.diffuse = diffuse[slot], st.specular = specular[slot];
    st.dcnt = diffuseCounter[slot], st.scnt = specularCounter[slot];
    st.dirty = 0u;
    return st;
}
SKH_DI void store_pix(const PixState& st, uint32_t slot, float4* __restrict__ accum, float4* __restrict__ diffuse, float4* __restrict__ specular,
                      uint16_t* __restrict__ diffuseCounter, uint16_t* __restrict__ specularCounter)
{
    if (st.dirty & 1u)
        accum[slot] = st.accum;
    if (st.dirty & 2u)
        diffuse[slot] = st.diffuse, diffuseCounter[slot] = (uint16_t)st.dcnt;
    if (st.dirty & 4u)
        specular[slot] = st.specular, specularCounter[slot] = (uint16_t)st.scnt;
}
// AOV + accumulation epilogue of one launch of `spl` samples at sub-frame index `subframeIndex` (OptixRender.cu:169-247)
SKH_DI float4 finalize_one(const FrameP& fp, uint32_t subframeIndex, uint32_t spl, v3 result, v3 dsum, v3 ssum, uint32_t diffuseSamples,
                           uint32_t specularSamples, PixState& st)
{
    const v3 exposure = mk3(fp.exposure[0], fp.exposure[1], fp.exposure[2]);
    result = result / (float)spl;
    float4 diffuseOut = make_float4(0.0f, 0.0f, 0.0f, 1.0f), specularOut = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    if (diffuseSamples > 0)
    {
        dsum = dsum / (float)diffuseSamples;
        const uint32_t prev = subframeIndex > 0 ? st.dcnt : 0u;
        const float4 h = st.diffuse;
        const v3 a = accumulate(mk3(h), dsum, exposure, prev);
        diffuseOut = make_float4(a.x, a.y, a.z, 1.0f);
        st.diffuse = diffuseOut;
        st.dcnt = (uint32_t)(uint16_t)(prev + diffuseSamples);
        st.dirty |= 2u;
    }
    else
    {
        if (subframeIndex == 0)
        {
            st.diffuse = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            st.dcnt = 0;
            st.dirty |= 2u;
        }
        diffuseOut = st.diffuse;
    }
    if (specularSamples > 0)
    {
        ssum = ssum / (float)specularSamples;
        const uint32_t prev = subframeIndex > 0 ? st.scnt : 0u;
        const float4 h = st.specular;
        const v3 a = accumulate(mk3(h), ssum, exposure, prev);
        specularOut = make_float4(a.x, a.y, a.z, 1.0f);
        st.specular = specularOut;
        st.scnt = (uint32_t)(uint16_t)(prev + specularSamples);
        st.dirty |= 4u;
    }
    else
    {
        if (subframeIndex == 0)
        {
            st.specular = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            st.scnt = 0;
            st.dirty |= 4u;
        }
        if (st.scnt > 0)
            specularOut = st.specular;
    }
    float4 out;
    if (fp.debug == 2)
        out = diffuseOut;
    else if (fp.debug == 3)
        out = specularOut;
    else if (fp.enableAccumulation && fp.debug == 0)
    {
        const float4 h = st.accum;
        const v3 a = accumulate(mk3(h), result, exposure, subframeIndex);
        out = make_float4(a.x, a.y, a.z, 1.0f);
        st.accum = out;
        st.dirty |= 1u;
    }
    else
        out = make_float4(result.x, result.y, result.z, 1.0f);
    return out;
}

__global__ void __launch_bounds__(256)
    k_finalize(FrameP fp, const uint32_t* __restrict__ tileXY, const float* __restrict__ sums, float4* __restrict__ accum,
               float4* __restrict__ diffuse, float4* __restrict__ specular, uint16_t* __restrict__ diffuseCounter,
               uint16_t* __restrict__ specularCounter, float4* __restrict__ image)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t px, py;
    if (slot >= fp.numSlots || !slot_to_pixel(fp, tileXY, slot, px, py))
        return;
    const size_t N = fp.numSlots;
    const v3 result = mk3(sums[slot], sums[slot + N], sums[slot + 2 * N]);
    const v3 dsum = mk3(sums[slot + 3 * N], sums[slot + 4 * N], sums[slot + 5 * N]);
    const v3 ssum = mk3(sums[slot + 6 * N], sums[slot + 7 * N], sums[slot + 8 * N]);
    PixState st = load_pix(slot, accum, diffuse, specular, diffuseCounter, specularCounter);
    const float4 out = finalize_one(fp, fp.subframeIndex, fp.samplesThisLaunch, result, dsum, ssum, (uint32_t)sums[slot + 9 * N],
                                    (uint32_t)sums[slot + 10 * N], st);
    store_pix(st, slot, accum, diffuse, specular, diffuseCounter, specularCounter);
    if (image)
        image[(size_t)py * fp.width + px] = out;
}

// batch of fp.batch sub-frames of ONE sample each, traced together: apply their accumulation steps in sub-frame order
// (the accumulator is an order-dependent LDR-space lerp: OptixRender.cu:60-78)
__global__ void __launch_bounds__(256)
    k_finalize_batch(FrameP fp, const uint32_t* __restrict__ tileXY, PathS ps, float4* __restrict__ accum, float4* __restrict__ diffuse,
                     float4* __restrict__ specular, uint16_t* __restrict__ diffuseCounter, uint16_t* __restrict__ specularCounter,
                     float4* __restrict__ image)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t px, py;
    if (slot >= fp.numSlots || !slot_to_pixel(fp, tileXY, slot, px, py))
        return;
    const size_t S = ps.stride;
    float4 out = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    PixState st = load_pix(slot, accum, diffuse, specular, diffuseCounter, specularCounter);
    for (uint32_t sub = fp.finalFirst; sub < fp.finalFirst + fp.finalCount; ++sub)
    {
        const size_t p = (size_t)sub * fp.numSlots + slot;
        const v3 rad = mk3(ps.rad()[p]);
        const uint32_t fe = (reinterpret_cast<const uint32_t*>(ps.base)[p + 7 * S] >> PF_EVENT_SHIFT) & 3u;
        // `result += prd.radiance` starts from 0.0f in the reference (OptixRender.cu:83,154): keep that addition
        const v3 result = mk3(0.0f) + rad;
        out = finalize_one(fp, fp.subframeIndex + sub, 1u, result, fe == 2 ? result : mk3(0.0f), fe == 3 ? result : mk3(0.0f), fe == 2 ? 1u : 0u,
                           fe == 3 ? 1u : 0u, st);
    }
    store_pix(st, slot, accum, diffuse, specular, diffuseCounter, specularCounter);
    if (image)
        image[(size_t)py * fp.width + px] = out;
}

// compact tile-major slots -> row-major W x H image (and back-end of the multi-GPU gather)
__global__ void __launch_bounds__(256) k_detile(const float4* __restrict__ src, const uint32_t* __restrict__ tileXY,
                                               uint32_t numSlots, uint32_t tileShift, uint32_t width, uint32_t height,
                                               float4* __restrict__ dst)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= numSlots)
        return;
    const uint32_t tile = slot >> (2 * tileShift);
    const uint32_t m = slot & ((1u << (2 * tileShift)) - 1u);
    const uint32_t px = tileXY[2 * tile] + compact1by1(m), py = tileXY[2 * tile + 1] + compact1by1(m >> 1);
    if (px < width && py < height)
        dst[(size_t)py * width + px] = src[slot];
}

__global__ void k_add_stats(const uint32_t* __restrict__ counts, uint32_t numBounces, StatsDev* __restrict__ stats)
{
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        unsigned long long r = 0, s = 0;
        for (uint32_t b = 0; b < numBounces; ++b)
            for (uint32_t g = 0; g < SKH_SHARDS; ++g)
            {
                r += counts[(2 * b * SKH_SHARDS + g) * SKH_COUNT_STRIDE];
                s += counts[((2 * b + 1) * SKH_SHARDS + g) * SKH_COUNT_STRIDE];
            }
        stats->raysRadiance += r;
        stats->raysShadow += s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Post: postprocessing/Tonemappers.cu:6-135 (Reinhard / ACES fitted / ACES film + gamma), in place
// ------------------------------------------------------------------------------------------------------------
SKH_DI v3 tm_reinhard(const v3& c)
{
    const float lum = dot(c, mk3(0.299f, 0.587f, 0.114f));
    return c / (lum + 1);
}
SKH_DI v3 tm_aces_film(const v3& x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    const v3 r = (x * (a * x + mk3(b))) / (x * (c * x + mk3(d)) + mk3(e));
    return mk3(saturatef(r.x), saturatef(r.y), saturatef(r.z));
}
SKH_DI v3 tm_aces_fitted(v3 color)
{
    const float in[9] = { 0.59719f, 0.35458f, 0.04823f, 0.07600f, 0.90834f, 0.01566f, 0.02840f, 0.13383f, 0.83777f };
    const float out[9] = { 1.60475f, -0.53108f, -0.07367f, -0.10208f, 1.10813f, -0.00605f, -0.00327f, -0.07276f, 1.07602f };
    v3 c = mk3(in[0] * color.x + in[1] * color.y + in[2] * color.z, in[3] * color.x + in[4] * color.y + in[5] * color.z,
               in[6] * color.x + in[7] * color.y + in[8] * color.z);
    const v3 a = c * (c + mk3(0.0245786f)) - mk3(0.000090537f);
    const v3 b = c * (0.983729f * c + mk3(0.4329510f)) + mk3(0.238081f);
    c = a / b;
    const v3 o = mk3(out[0] * c.x + out[1] * c.y + out[2] * c.z, out[3] * c.x + out[4] * c.y + out[5] * c.z,
                     out[6] * c.x + out[7] * c.y + out[8] * c.z);
    return mk3(saturatef(o.x), saturatef(o.y), saturatef(o.z));
}
__global__ void __launch_bounds__(256) k_tonemap(float4* __restrict__ image, uint32_t n, uint32_t type, float ex, float ey, float ez,
                                                float gamma)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float4 px = image[i];
    v3 c = mk3(px.x, px.y, px.z);
    if (type >= 1 && type <= 3)
    {
        const v3 r = c * mk3(ex, ey, ez);
        c = type == 1 ? tm_reinhard(r) : (type == 2 ? tm_aces_fitted(r) : tm_aces_film(r));
        px.w = 1.0f;
    }
    if (gamma > 0.0f)
    {
        const float g = 1.0f / gamma;
        c = mk3(skm::powf_(c.x, g), skm::powf_(c.y, g), skm::powf_(c.z, g));
        px.w = 1.0f;
    }
    image[i] = make_float4(c.x, c.y, c.z, px.w);
}

// instance records: transform the BLAS root box into world space (TLAS primitive box) and fill DevInstance
__global__ void k_instance_boxes(const HostInstance* __restrict__ instances, const float* __restrict__ w2oAll /*12 per inst*/,
                                 const uint8_t* __restrict__ validAll, const float* __restrict__ triGroupBounds,
                                 const int* __restrict__ triGroupRoot, const float* __restrict__ segGroupBounds,
                                 const int* __restrict__ segGroupRoot, uint32_t nMeshes, uint32_t nCurves, uint32_t n,
                                 DevInstance* __restrict__ out, float4* __restrict__ boxLo, float4* __restrict__ boxHi,
                                 uint32_t* __restrict__ grp)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const HostInstance in = instances[i];
    DevInstance d;
#pragma unroll
    for (int k = 0; k < 12; ++k)
        d.w2o[k] = w2oAll[12 * (size_t)i + k];
    d.type = in.type;
    d.pad = 0;
    d.mask = in.type == 0 ? 1u : (in.type == 2 ? 2u : 4u);
    const bool isCurve = in.type == 2;
    const bool has = isCurve ? (in.geom < nCurves) : (in.geom < nMeshes);
    int root = SKH_REF_INVALID;
    const float* gb = nullptr;
    if (has)
    {
        root = isCurve ? segGroupRoot[in.geom] : triGroupRoot[in.geom];
        gb = (isCurve ? segGroupBounds : triGroupBounds) + 6 * (size_t)in.geom;
    }
    bool valid = validAll[i] != 0 && root != SKH_REF_INVALID;
    d.rootRef = root;
    float4 lo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), hi = lo;
    if (valid)
    {
        v3 wlo = mk3(INFINITY), whi = mk3(-INFINITY);
#pragma unroll
        for (int k = 0; k < 8; ++k)
        {
            const v3 p = mk3((k & 1) ? gb[3] : gb[0], (k & 2) ? gb[4] : gb[1], (k & 4) ? gb[5] : gb[2]);
            const v3 w = xform_point(in.o2w, p);
            wlo = mk3(fminf(wlo.x, w.x), fminf(wlo.y, w.y), fminf(wlo.z, w.z));
            whi = mk3(fmaxf(whi.x, w.x), fmaxf(whi.y, w.y), fmaxf(whi.z, w.z));
        }
        lo = make_float4(wlo.x, wlo.y, wlo.z, 0.0f);
        hi = make_float4(whi.x, whi.y, whi.z, 0.0f);
        // the BLAS boxes are inflated by 2^-20 relative; cover that and the transform rounding
        inflate_box(lo, hi);
        inflate_box(lo, hi);
    }
    else
        d.mask = 0;
    out[i] = d;
    boxLo[i] = lo;
    boxHi[i] = hi;
    grp[i] = 0;
}

// TLAS build on the device: the primitives of the top level are the valid instances (list made on the host from what it
// already knows: a finite inverse transform, a non-empty BLAS); their boxes and traversal records, one per leaf
__global__ void k_tlas_leaves(const uint32_t* __restrict__ leafInst, uint32_t n, const DevInstance* __restrict__ inst, const float4* __restrict__ boxLo,
                              const float4* __restrict__ boxHi, float4* __restrict__ leafLo, float4* __restrict__ leafHi, uint32_t* __restrict__ grp,
                              DevInstance* __restrict__ out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n)
        return;
    const uint32_t i = leafInst[r];
    leafLo[r] = boxLo[i];
    leafHi[r] = boxHi[i];
    grp[r] = 0u;
    DevInstance d = inst[i];
    d.pad = i; // the instance this leaf belongs to (hit records report it)
    out[r] = d;
}
__global__ void k_permute_instances(const DevInstance* __restrict__ src, const uint32_t* __restrict__ order, uint32_t n, DevInstance* __restrict__ dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] = src[order[i]];
}

// Tight world box of a mesh instance: the box of its transformed VERTICES instead of the box of the eight transformed corners
// of the object-space box (for a round object turned by 45 degrees the latter has twice the footprint).  One workgroup per
// instance; the result only ever SHRINKS the corner box (component-wise intersection) and is padded like it.  Instances of
// very large meshes keep the corner box (maxVerts), curve instances too.
__global__ void __launch_bounds__(256) k_instance_tight_boxes(const HostInstance* __restrict__ instances, const DevInstance* __restrict__ dev,
                                                              const uint4* __restrict__ meshes, const uint8_t* __restrict__ verts, uint32_t nMeshes,
                                                              uint32_t maxVerts, float4* __restrict__ boxLo, float4* __restrict__ boxHi)
{
    __shared__ float s_red[6][256];
    const uint32_t i = blockIdx.x;
    const HostInstance in = instances[i];
    if (in.type == 2 || in.geom >= nMeshes || dev[i].mask == 0)
        return;
    const uint4 me = meshes[in.geom];
    if (me.w == 0 || me.w > maxVerts)
        return;
    v3 lo = mk3(INFINITY), hi = mk3(-INFINITY);
    for (uint32_t v = threadIdx.x; v < me.w; v += blockDim.x)
    {
        const float4 p = *reinterpret_cast<const float4*>(verts + (size_t)(me.z + v) * 32);
        const v3 w = xform_point(in.o2w, mk3(p.x, p.y, p.z));
        lo = mk3(fminf(lo.x, w.x), fminf(lo.y, w.y), fminf(lo.z, w.z));
        hi = mk3(fmaxf(hi.x, w.x), fmaxf(hi.y, w.y), fmaxf(hi.z, w.z));
    }
    s_red[0][threadIdx.x] = lo.x, s_red[1][threadIdx.x] = lo.y, s_red[2][threadIdx.x] = lo.z;
    s_red[3][threadIdx.x] = hi.x, s_red[4][threadIdx.x] = hi.y, s_red[5][threadIdx.x] = hi.z;
    __syncthreads();
    for (uint32_t st = 128; st >= 1; st >>= 1)
    {
        if (threadIdx.x < st)
            for (int k = 0; k < 6; ++k)
                s_red[k][threadIdx.x] = k < 3 ? fminf(s_red[k][threadIdx.x], s_red[k][threadIdx.x + st]) : fmaxf(s_red[k][threadIdx.x], s_red[k][threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        float4 tlo = make_float4(s_red[0][0], s_red[1][0], s_red[2][0], 0.0f), thi = make_float4(s_red[3][0], s_red[4][0], s_red[5][0], 0.0f);
        // same padding as the corner box (BLAS boxes inflated by 2^-20 relative, transform rounding), once more for the
        // difference between the forward transform used here and the inverse the traversal applies to the ray
        inflate_box(tlo, thi);
        inflate_box(tlo, thi);
        inflate_box(tlo, thi);
        const float4 clo = boxLo[i], chi = boxHi[i];
        boxLo[i] = make_float4(fmaxf(clo.x, tlo.x), fmaxf(clo.y, tlo.y), fmaxf(clo.z, tlo.z), 0.0f);
        boxHi[i] = make_float4(fminf(chi.x, thi.x), fminf(chi.y, thi.y), fminf(chi.z, thi.z), 0.0f);
    }
}

} // namespace skh
