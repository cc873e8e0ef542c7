// strelka_hip -- host side of the C ABI declared in include/strelka_hip.h (gfx950 / ROCm only).
// One translation unit: device code comes from skh_device.h / skh_bvh.h / skh_kernels.h.
#include "../../include/strelka_hip.h"
#include "skh_kernels.h"

#include <dlfcn.h>
// RCCL is dlopen()ed on first use (skh_comm_*); the handful of types its point-to-point API needs are declared here, so that a
// box without the RCCL development headers still builds the renderer (nccl.h 2.x: the layouts below are part of its stable ABI)
typedef struct ncclComm* ncclComm_t;
typedef struct
{
    char internal[128];
} ncclUniqueId;
typedef enum
{
    ncclSuccess = 0
} ncclResult_t;
typedef enum
{
    ncclFloat = 7 // ncclFloat32
} ncclDataType_t;

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace skh;

static_assert(sizeof(skh_vertex) == 32 && sizeof(skh_instance) == 64 && sizeof(skh_light) == 112, "ABI layout");
static_assert(sizeof(skh_material) == 64 && sizeof(skh_frame_params) == 176 && sizeof(skh_ray) == 32, "ABI layout");
static_assert(sizeof(skh_hit) == 20 && sizeof(Light) == 112 && sizeof(Material) == 64 && sizeof(HostInstance) == 64, "layout");

namespace
{
struct DevBuf
{
    void* p = nullptr;
    size_t bytes = 0;
    template <typename T>
    T* as() const
    {
        return reinterpret_cast<T*>(p);
    }
};

enum KernelClass
{
    KC_TRACE_CLOSEST = 0,
    KC_TRACE_SHADOW,
    KC_SHADE,
    KC_RAYGEN,
    KC_ACCUM,
    KC_SORT,
    KC_COUNT
};

#ifndef SKH_RI_MAX_LEVELS
#define SKH_RI_MAX_LEVELS 1024 // per-level counters of the reinsertion pass's refit: a ring (a deeper tree wraps around it; tests build a variant with 3)
#endif
#define SKH_MAX_LAUNCH_ROUNDS 140 // MAX_BOUNCES (128) + slack

struct TimedSpan
{
    int cls;
    hipEvent_t a, b;
};
} // namespace

struct skh_context
{
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr; // any-hit launches when `overlap` is on: shadow[b] runs beside closest[b+1] and fills its tail
    int overlap = 1; // 0 off, 1 for small passes only (<= 8 M paths: the interactive one-sub-frame-per-call mode, +7 %), 2 always
    hipEvent_t evShade = nullptr, evShadow = nullptr;
    ncclComm_t comm = nullptr; // multi-GPU tile gather (skh_comm_init)
    int commWorld = 1, commRank = 0;
    DevBuf dTileSend;
    uint32_t* hOverflow = nullptr; // pinned, device-visible: traversal-stack overflow flag (DevScene::overflowFlag)
    uint32_t stackOverflows = 0; // calls that failed with it since the last skh_reset_stats
    std::string err;
    int numCUs = 256;

    // scene (host copies needed for the build)
    std::vector<skh_mesh> meshes;
    std::vector<skh_curve> curves;
    std::vector<uint32_t> curveVertexCounts;
    uint32_t nVerts = 0, nIndices = 0, nPoints = 0, nInstances = 0, nLights = 0, nMaterials = 0;
    std::vector<skh_instance> instances;

    DevBuf dVerts, dIndices, dMeshes, dPoints, dRadii, dInstances, dLights, dMaterials, dHairConst;
    DevBuf dShadeTris, dShadeInst; // shading side: de-indexed triangle records, instance records that carry their mesh's base
    DevBuf dCurveSegBase, dSegStartAll;
    // accel
    DevBuf dTexels, dTexDesc, dScatterXY, dRaygenBase;
    uint32_t raygenBlocksPerSub = 0, raygenValidPerSub = 0;
    uint32_t nTextures = 0;
    bool hasHairMaterial = false; // selects the k_shade build that carries df::chiang_hair_bsdf
    // Speculative sub-frame batching for the reference's call pattern (one render() per sub-frame, RenderPass.cpp:441-447): once two
    // consecutive calls continue the same frame (same parameters, subframe_index + 1), the next call traces several sub-frames
    // ahead in ONE wavefront pass and the calls after it only apply their accumulation step to the radiances already in the path
    // state.  Exact (sub-frame batching is exact); a call that does not continue the frame discards what is left.
    struct Speculation
    {
        bool valid = false;
        skh_frame_params params; // of the pass (subframe_index = its first sub-frame)
        uint32_t count = 0, consumed = 0, lastBatch = 1, streak = 0;
        skh_frame_params last; // the previous call's parameters
        bool haveLast = false;
        uint64_t passRadiance = 0, passShadow = 0; // rays the pass traced (all `count` sub-frames; counted when traced)
        // option speculate_async: the pass AFTER the one being consumed is already being traced -- on c->stream, into the other
        // path-state buffer -- while the caller collects this pass's sub-frames (accumulation steps and map() copies on stream3)
        uint32_t buf = 0; // path-state buffer of the pass being consumed (0 = dPath, 1 = dPathB)
        bool nextInFlight = false;
        skh_frame_params nextParams;
        uint32_t nextCount = 0;
        unsigned long long statsMark[2] = { 0, 0 }; // dStats {raysRadiance, raysShadow} when the pass in flight was launched
    } spec;
    uint32_t speculateAsync = 1; // option speculate_async: 224 -> 180 ms per frame in the reference caller's loop with map() (docs/LOG.md)
    hipStream_t stream3 = nullptr; // accumulation steps + image copies beside a pass in flight (non-blocking: no implicit sync with the null stream)
    DevBuf dPathB;
    uint32_t pathBStride = 0;
    // sub-frames traced ahead and then thrown away (camera move, a setter, resize): their rays are taken out of the ray counts again,
    // so that a Mray/s figure from skh_get_stats only counts rays whose sub-frame was delivered
    uint64_t discardedRadiance = 0, discardedShadow = 0;
    uint32_t discardedSubframes = 0;
    uint32_t speculateGrow = 2; // option speculate_grow: how fast the look-ahead grows while the caller keeps continuing a frame (x2 per pass: 2, 4, 8; 8 = straight to the cap at the second call)
    uint32_t speculateMax = 8; // option "speculate": most sub-frames traced ahead in one pass (0 / 1 = off).  8: a pass stays below ~25 ms at 1080p
    DevBuf dTriNodes, dTris, dSegNodes, dSegs, dTlasNodes, dTlasInst, dDevInst, dTravInst;
    int tlasRoot = SKH_REF_INVALID;
    bool accelBuilt = false;
    // skh_refit_accel: what the last build leaves behind for it -- the triangle tree's leaf order and primitive tables (k_gather_tris' inputs), the
    // nodes' levels -- and what must not have changed since (the signature of the mesh table + index buffer, the vertex count)
    bool refitReady = false;
    DevBuf dTriOrder, dTriMeshK, dTriLocalK, dWInstK, dWFirstK, dTriNodeBox;
    std::vector<uint32_t> triLevelStart;
    uint32_t nMeshTrisBuilt = 0, nWInstBuilt = 0, triNumNodes = 0, lastBuildFlags = 0, nMeshGroupsBuilt = 0, nGroup1Built = 0;
    uint64_t geomSig = 0, builtGeomSig = 0;
    uint32_t builtNVerts = 0;
    double msRefit = 0.0;
    uint32_t refits = 0;
    // ... and of the curve build: k_gather_segs' tables, the curve tree's levels, the signature of the curve sets' topology (control points and radii may change)
    DevBuf dSegOrder, dSegBuildStartK, dSegLocalK, dSegInstOfK, dSegNodeBox;
    std::vector<uint32_t> segLevelStart;
    uint32_t nSubBuilt = 0, segNumNodes = 0;
    uint64_t curveSig = 0, builtCurveSig = 0;
    bool curvePointsEdited = false, curveRefitReady = false;
    uint32_t nTris = 0, nSegs = 0;

    // frame
    uint32_t width = 0, height = 0, tileSize = 32, tileShift = 5, numTiles = 0, numSlots = 0;
    std::vector<uint32_t> tileXY;
    bool customTiles = false;
    DevBuf dTileXY, dAccum, dDiffuse, dSpecular, dDiffCnt, dSpecCnt, dSums, dPath, dRayQ[2], dHits, dShadowQ, dContrib, dCounts,
        dOvf, dOvf2, dStats, dScratchImage;
    uint32_t traceBlocks = 0;
    uint32_t curveSplitBuilt = 1;
    uint32_t numWorldCurves = 0; // curve instances under identity transforms that the world-only kernel walks itself (skh_build_accel)
    int worldCurveRoot[SKH_WORLD_CURVES];
    uint32_t worldCurveInst[SKH_WORLD_CURVES];
    uint32_t worldCurveMerged = 0; // bit k: table entry k is a merged transform group (its hits take their instance from the segment record)
    uint32_t worldCurveIdentLast = 0;
    uint32_t hierNodes = 0;  // 4-wide nodes of the triangle and curve trees (skh_build_accel): decides the automatic fetch_chunk
    int32_t fetchChunk = -1; // option fetch_chunk: queue positions a trace wave reserves per atomic; 0 = one atomic per refill; -1 = automatic
    uint32_t numTlasLeaves = 0;
    bool countTraversal = false, timing = false;
    // scheduling of the persistent trace kernels, measured on MI355X (kitchen C3, 32 sub-frames per pass):
    uint32_t wavesPerCU = 28; // resident waves per CU (7 per SIMD at <= 72 VGPRs; the curve build is resident 16 at a time whatever is asked: 128 VGPRs)
    uint32_t wavesPerCUShadow = 28; // the any-hit build of the triangle kernel fits 7 per SIMD
    uint32_t wavesPerCUShadowWorld = 32; // ... its world-only build 8 (SKH_WORLD_ANYHIT_MIN_WAVES)
    uint32_t queueConstBits[2] = { 0, 0 }; // materialTmin / shadowTmin the constant planes of the ray queues hold (bit patterns)
    bool queueConstFilled = false; // ... and whether they hold them at all (alloc_frame resets it)
    uint32_t reinsertRounds = 8; // option reinsert_rounds: rounds of parallel reinsertion over the PLOC tree of the triangle build (skh_bvh.h k_ri_*)
    uint32_t reinsertCurveRounds = 4; // option reinsert_curve_rounds: the same pass over the curve sub-segment trees
    uint32_t reinsertMinSize = 0; // option reinsert_min_size: 0 = auto (1 up to 4 M primitives, 32 beyond)
    skh_build_info buildInfo = {};
    uint32_t plocTop = 0; // triangle build: clusters left at which PLOC switches to the wide neighbour search (option ploc_top; 0 = never)
    uint32_t wavesPerCUWorld = 32; // the world-only closest-hit build: 64 VGPRs, 8 per SIMD (SKH_WORLD_CLOSEST_MIN_WAVES)
    uint32_t smallWavesClosest = 0, smallWavesShadow = 0; // (0 = automatic: 20 / 12 for triangle scenes, 16 / 16 with curves -- there the any-hit launch is the heavier one: hair 1-spp calls 687 against 664 Mray/s) overlapped (small) passes: waves per CU of each of the two concurrent trace kernels (0 = wavesPerCU); 16/16: +4 % on 1-spp 1080p launches over 24/24; round 6: the closest-hit launch is the one on the critical path -- 20/12: 1-spp 1080p calls 3.49 -> 3.39 ms, drop-in +1 % (18/14 3.41, 22/10 3.48, 24/8 3.67: then the any-hit launch is the long one)
    uint32_t gridOverride = 0; // set by render_one around its launches
    uint32_t fetchMinClosestSmall = 48; // option fetch_min_closest_small
    bool fetchMinClosestSet = false;
    uint32_t smallWavesFirst = 0, smallWavesLast = 0; // (small overlapped passes) waves per CU of the FIRST closest-hit launch and of the LAST any-hit launch, which have the machine to themselves; 0 = as the others, except that the last any-hit launch of a triangle scene takes 20 of 32 instead of 12 (1-spp 1080p call 3.13 -> 3.10 ms; 16 / 24 the same, 32: 3.13; the first closest-hit launch at 24 / 28 / 32: 3.12 / 3.13 / 3.15)
    int tailSplit = 1; // option tail_split: 1 (default) = the world-only triangle kernels' SPLIT build for every launch of a scene whose hierarchy has more than 16 384 nodes -- once a wave finds the ray queue dry (the tail
                       // phase of k_trace: skh_trace_body.inc included a second time), its idle lanes take stack entries of the lanes that still hold a ray --, 2 = for every launch whatever the hierarchy (tests), -1 = for passes
                       // of 2^17 ... 2^23 paths only, 0 = never.  Per launch 285 + 274.5 n -> 225 + 272.9 n us (closest-hit, n sub-frames of 2.07 M paths), 228 + 115.0 n -> 148 + 114.8 n (any-hit): the main phase is the plain
                       // build's, instruction for instruction.  Hierarchies that fit the L2 many times over (Cornell: 30 triangles, rays of two or three steps) have no tails to speak of and lose by the hand-out:
                       // closest-hit 27.6 -> 28.4 ms with it (the same scenes whose launches are bound by the queue cursors: fetch_chunk)
    bool splitNow = false; // set by render_one around its launches
    uint32_t fetchMinClosest = 32 /* 24 until round 5: on the reinserted trees 32 is 0.4 ... 0.9 % ahead on all three kitchens, gpurun_out/r6c */, fetchMinShadow = 48; // idle lanes before a wave pulls new rays from the queue (round 3, world-space hierarchy: any-hit 32 -> 48: 39.7 -> 36.2 ms, 56: 37.5, 64: 51; closest 12..40 within 1 %)
    uint32_t curveFetchMinClosest = 16, curveFetchMinShadow = 16 /* 24 until the result writes got cheaper (round 5, late): hair any-hit 47.5-47.8 -> 46.5-47.0 ms with 12 ... 20, gpurun_out/r7v */, curveNodeBreakClosest = 20, curveNodeBreakShadow = 20; // the same four for the curve build (6 waves/SIMD): hair 482 vs 465 Mray/s
    uint32_t nodeBreakClosest = 32, nodeBreakShadow = 28; // (closest: 24 -> 32 in round 3 for the world-only kernel: kitchen 86.6 -> 85.9 ms, unshared 74.7 -> 73.6, three runs each;
                                                          // shadow: 20 -> 28 in round 4, with the shared triangle pass: kitchen 34.85 -> 34.3 ms, unshared 29.3 -> 28.55, three runs each; 36: 34.35 / 28.7)
    // leave the node loop when fewer than x/64 of the wave's rays are still descending
    uint32_t worldCurveMin = 32; // ... of the world-only kernel with the curve block (option curve_min sets both)
    uint32_t curveMin = 48; // (cooperative curve block) end-point runs queued by the parked lanes before the block runs, one run per lane (round 3, Mray/s on the hair
                            // stand-in: 16: 712, 32: 907, 40: 958, 48: 984, 56: 971, 64: 887; round 2 counted parked LANES whose owners ran their own runs: 48: 334)
    uint32_t leafMin = 16; // postpone the minority kind of leaf work unless it has this many lanes (0 = never postpone; measured +1.5 % at 16)
    uint32_t subframeBatch = 0, batchCapacity = 1; // option subframe_batch: 0 = auto
    uint32_t queueRegion = 64; // positions per queue shard (RayQ::region): the queues hold SKH_SHARDS * queueRegion rays
    bool tightInstanceBoxes = true; // TLAS leaf boxes from the transformed vertices, not from the transformed object box
    uint32_t curveLeaf = 1; // sub-segments per curve leaf (option curve_leaf; hair stand-in after the intersector's early exit, Mray/s: 1: 1456, 2: 1392, 3: 1309, 4: 1240;
                            // the cooperative block takes two candidates per lane and block)
    uint32_t numMergedCurveInst = 0; // (build result) curve instances that share the merged world-space tree
    uint32_t splitPairs = 0; // (option split_pairs, tenths) triangle trees: a two-triangle subtree whose box exceeds this x the summed areas of its triangles' boxes may be opened into two one-triangle leaves (0 = off)
    uint32_t curveMerge = 1; // (option curve_merge) curve instances under identity transforms share ONE world-space tree in the world-only curve kernel
    uint32_t curveSegNode = 0; // (option curve_segnode) 1: the curve tree is built over whole segments and ends in SEGMENT NODES (skh_bvh.h k_segnode_emit): a segment is a
                               // candidate at most once per ray; 0: parameter sub-ranges as primitives (curve_split), rounds 3-5
    uint32_t curveStrandMajor = 0; // (option curve_strand_major, segment-node build) 1: leaf records and segment nodes at the segment's own index (consecutive segments of a strand adjacent)
    uint32_t curveSplit = 4; // parameter sub-ranges per curve segment in the curve BLAS (round 3, one-sub-segment leaves: 2: 1460, 3: 1480, 4: 1503 Mray/s; build time and leaf
                             // memory grow with it.  Round 2, ms per 1080p sub-frame: 1: 61.6, 2: 52.1, 4: 49.7, 8: 50.1)
    // TLAS builder.  1 (default): on the GPU -- PLOC over the instance boxes with a 96-neighbour search, the BLAS builder, no host round
    // trip: 4 / 5 / 9 ms for 2 k / 20 k / 100 k instances.  0: exact three-axis sweep SAH on the host, O(n log^2 n) single-threaded
    // (4 / 45 ms for 2 k / 20 k), whose tree enters 5 % fewer instances (1.28 vs 1.35 per ray on the kitchen stand-in with only its
    // room baked: closest-hit 97.6 vs 100.3 ms; PLOC radius 24 .. 512 makes no difference, docs/LOG.md).  With bake_world 4 a top level
    // only exists for curve sets, for light proxies beside them, and for scenes with more than bake_budget_mtris instanced triangles:
    // every structure the bench workloads traverse is built by GPU kernels.  2: the sweep up to 8192 TLAS leaves, the GPU beyond.
    uint32_t tlasBuild = 1;
    uint32_t tlasOpen = 1; // TLAS opening: up to tlasOpen x numInstances leaves; 1 = one leaf per instance (default: on the kitchen stand-in 2..16 were 4-9 % slower, more instance entries for no fewer nodes)
    // bake_world: mesh instances that skip the TLAS -- their triangles are carried to world space once and join ONE extra
    // group of the triangle build that every ray walks first, with no instance entry (DESIGN.md section 2 "bake_world").
    // 0 off; 1: instances whose mesh has a single user (what HdStrelka's per-instance meshes are, RenderPass.cpp:126-129,252-257);
    // 2: also instances of meshes with <= bakeSmallTris triangles (room shells, boards, quads: big boxes that every ray enters
    // for a dozen triangles), while they add at most max(unique triangles, 2^20) triangles
    uint32_t bakeWorld = 4, bakeSmallTris = 64, bakeBudgetMTris = 64;
    bool worldKernel = true; // option world_kernel: scenes with an empty top level run the world-only build of k_trace (0 = the general build: A/B, tests)
    std::vector<uint8_t> baked; // per instance, valid after skh_build_accel
    int worldRoot = SKH_REF_INVALID, lightRoot = SKH_REF_INVALID; // roots of the two baked groups (mesh instances, light proxies) inside dTriNodes
    uint32_t nBakedTris = 0, nBakedInst = 0;
    uint32_t mortonBits = 10; // per axis, in the builders' sort keys (option morton_bits 4..21: 10 / 13 / 16 / 20 measured within the +-1.5 % the tree's shape varies by anyway)
    uint32_t leafLines = 0;   // 1: triangle leaves laid out by 128-byte line (skh_bvh.h: k_leaf_place): -11 % fetched lines, same time (docs/LOG.md)
    uint32_t nTriSlots = 0;
    LightBox lightBox = { { -INFINITY, -INFINITY, -INFINITY }, { INFINITY, INFINITY, INFINITY } }; // around the baked light proxies' group (skh_build_accel), with the node encoder's margin
    uint32_t mergeLightProxies = 0; // option merge_light_proxies: baked light proxies share the world-space mesh triangles' tree (any-hit queries skip their triangles) instead of a tree of their own that every radiance ray visits (a measured loss: docs/LOG.md round 5)
    uint32_t nShadeRecords = 0; // de-indexed shading triangle records (build_shading_tables)
    int32_t directRecordsOpt = -1; // option direct_records: -1 = by the counts (below), 0 / 1 forced
    uint32_t directRecords = 1; // baked mesh triangles name their shading record (SKH_PRIM_DIRECT); 0 when only (instance, mesh-local primitive) fits a 16-byte hit record
    uint32_t hitPrimRange = 0;  // primitive words of this scene's hits stay below it (records, or mesh-local indices; curve segments)
    uint32_t compactHits = 1;   // option compact_hits: 16-byte hit records in the render passes of world-only triangle scenes that fit (HitQ::primBits)
    uint32_t leafMaxTris = 2; // measured on MI355X: 2 beats 1, 3, 4, 6, 8 (the kernel is ALU bound, wasted triangle tests cost more than extra nodes)
    uint32_t buildQuality = 1; // 0: Karras radix tree (fastest build), 1: PLOC clustering (SAH-class quality)
    float sceneLo[3] = { 0, 0, 0 }, sceneHi[3] = { 1, 1, 1 };

    // timing
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> eventPool;
    size_t eventsUsed = 0;
    double msClass[KC_COUNT] = { 0, 0, 0, 0, 0, 0 };
    uint32_t launches[KC_COUNT] = { 0, 0, 0, 0, 0, 0 };
    double msBuild = 0.0;
};

#define SKH_TRY(ctx, expr)                                                                                              \
    do                                                                                                                  \
    {                                                                                                                   \
        hipError_t _e = (expr);                                                                                         \
        if (_e != hipSuccess)                                                                                           \
        {                                                                                                               \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                                            \
            return _e == hipErrorOutOfMemory ? SKH_OUT_OF_MEMORY : SKH_FAIL;                                           \
        }                                                                                                               \
    } while (0)

static skh_status dev_alloc(skh_context* c, DevBuf& b, size_t bytes)
{
    if (b.p && b.bytes >= bytes && b.bytes <= bytes * 2 + 4096)
        return SKH_OK;
    if (b.p)
    {
        (void)hipFree(b.p);
        b.p = nullptr;
        b.bytes = 0;
    }
    if (bytes == 0)
        bytes = 16;
    SKH_TRY(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return SKH_OK;
}
static skh_status dev_upload(skh_context* c, DevBuf& b, const void* src, size_t bytes)
{
    skh_status s = dev_alloc(c, b, bytes);
    if (s != SKH_OK)
        return s;
    if (bytes)
        SKH_TRY(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    return SKH_OK;
}
static void dev_free(DevBuf& b)
{
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

// Sobol generator matrices for the 5 dimensions the reference tabulates (RandomSampler.h:139-164), produced from the
// Joe-Kuo recurrences (d=2: s=1 a=0 m={1}; d=3: s=2 a=1 m={1,3}; d=4: s=3 a=1 m={1,3,1}; d=5: s=3 a=2 m={1,1,1}).
static void init_sobol_table(uint32_t tab[5][32])
{
    for (int i = 0; i < 32; ++i)
        tab[0][i] = 1u << (31 - i);
    const int S[4] = { 1, 2, 3, 3 };
    const uint32_t A[4] = { 0, 1, 1, 2 };
    const uint32_t M[4][3] = { { 1, 0, 0 }, { 1, 3, 0 }, { 1, 3, 1 }, { 1, 1, 1 } };
    for (int d = 0; d < 4; ++d)
    {
        uint32_t* v = tab[d + 1];
        const int s = S[d];
        for (int i = 0; i < s; ++i)
            v[i] = M[d][i] << (31 - i);
        for (int i = s; i < 32; ++i)
        {
            uint32_t x = v[i - s] ^ (v[i - s] >> s);
            for (int k = 1; k < s; ++k)
                if ((A[d] >> (s - 1 - k)) & 1u)
                    x ^= v[i - k];
            v[i] = x;
        }
    }
}

// world->object of an affine 3x4: the 3x3 by the fp64 adjugate, one rounding to fp32; the fourth column carries the object-to-world
// TRANSLATION itself (points go through xform_point_rel: R^-1 (p - T); DESIGN.md "instance transforms")
static bool invert_affine(const float* m, float* out)
{
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double tx = m[3], ty = m[7], tz = m[11];
    const double A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
    const double D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
    const double G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
    const double det = a * A + b * D + c * G;
    const double r = 1.0 / det;
    const double i00 = A * r, i01 = B * r, i02 = C * r, i10 = D * r, i11 = E * r, i12 = F * r, i20 = G * r, i21 = H * r, i22 = I * r;
#if SKH_ENTRY_REL
    out[0] = (float)i00, out[1] = (float)i01, out[2] = (float)i02, out[3] = (float)tx;
    out[4] = (float)i10, out[5] = (float)i11, out[6] = (float)i12, out[7] = (float)ty;
    out[8] = (float)i20, out[9] = (float)i21, out[10] = (float)i22, out[11] = (float)tz;
#else
    out[0] = (float)i00, out[1] = (float)i01, out[2] = (float)i02, out[3] = (float)(-(i00 * tx + i01 * ty + i02 * tz));
    out[4] = (float)i10, out[5] = (float)i11, out[6] = (float)i12, out[7] = (float)(-(i10 * tx + i11 * ty + i12 * tz));
    out[8] = (float)i20, out[9] = (float)i21, out[10] = (float)i22, out[11] = (float)(-(i20 * tx + i21 * ty + i22 * tz));
#endif
    bool finite = true;
    for (int k = 0; k < 12; ++k)
        finite = finite && std::isfinite(out[k]);
    return finite;
}

// ---------------------------------------------------------------------------------------------------------------
// generic LBVH build over n primitives with boxes + group ids already on the device
// ---------------------------------------------------------------------------------------------------------------
struct LbvhOut
{
    DevBuf nodes, sortedVals, groupRoot, groupBounds;
    std::vector<int> hostGroupRoot;
    uint32_t numNodes = 0;
    std::vector<uint32_t> levelStart; // the collapse's node slots per level: level L = [levelStart[L], levelStart[L + 1]) -- every child node lies in a later level (skh_refit_accel)
    uint32_t numSlots = 0; // entries of sortedVals: n, or more when the leaves were laid out by 128-byte line (0xffffffff = padding slot)
    // the reinsertion pass (skh_bvh.h k_ri_*), when it ran: rounds done, moves applied, sum of the internal nodes' box areas before / after, time
    uint32_t riRounds = 0, riMoves = 0, riMinSize = 0;
    double riCostBefore = 0.0, riCostAfter = 0.0, riMs = 0.0;
};

static skh_status lbvh_build(skh_context* c, uint32_t n, uint32_t nGroups, const std::vector<uint32_t>& groupCount,
                             const float4* dBoxLo, const float4* dBoxHi, const uint32_t* dGrp, int leafMax, bool ploc, LbvhOut& out,
                             int search = 0 /* PLOC search radius: 0 triangles, 1 TLAS, 2 curve sub-segments */, uint32_t lineRecBytes = 0 /* > 0: leaf records of this size, laid out by 128-byte line */,
                             uint32_t riRounds = 0 /* reinsertion rounds after PLOC */, uint32_t riMinSize = 1)
{
    hipStream_t st = c->stream;
    skh_status s;
    out.numSlots = n;
    if ((s = dev_alloc(c, out.groupBounds, sizeof(float) * 6 * (size_t)std::max(1u, nGroups))) != SKH_OK)
        return s;
    if ((s = dev_alloc(c, out.groupRoot, sizeof(int) * (size_t)std::max(1u, nGroups))) != SKH_OK)
        return s;
    if ((s = dev_alloc(c, out.nodes, sizeof(Node4) * ((size_t)std::max(1u, n) + 1))) != SKH_OK)
        return s;
    if ((s = dev_alloc(c, out.sortedVals, sizeof(uint32_t) * (size_t)std::max(1u, n))) != SKH_OK)
        return s;
    std::vector<uint32_t> groupFirst(nGroups);
    out.hostGroupRoot.assign(nGroups, SKH_REF_INVALID);
    uint32_t acc = 0;
    for (uint32_t g = 0; g < nGroups; ++g)
    {
        groupFirst[g] = acc;
        if (groupCount[g] > 0 && groupCount[g] <= (uint32_t)leafMax)
            out.hostGroupRoot[g] = ~(int)((acc << 3) | (groupCount[g] - 1u));
        acc += groupCount[g];
    }
    if (n == 0 || nGroups == 0)
    {
        if (nGroups)
            SKH_TRY(c, hipMemcpyAsync(out.groupRoot.p, out.hostGroupRoot.data(), sizeof(int) * nGroups, hipMemcpyHostToDevice, st));
        SKH_TRY(c, hipStreamSynchronize(st));
        return SKH_OK;
    }
    DevBuf gbU, keysA, keysB, valsB, hist, histSums, childL, childR, parent, rangeF, rangeL, flags, nodeLo, nodeHi, gFirst, gCount;
    auto cleanup = [&]() {
        for (DevBuf* b : { &gbU, &keysA, &keysB, &valsB, &hist, &histSums, &childL, &childR, &parent, &rangeF, &rangeL, &flags, &nodeLo, &nodeHi,
                           &gFirst, &gCount })
            dev_free(*b);
    };
#define LB_ALLOC(buf, bytes)                         \
    if ((s = dev_alloc(c, buf, (bytes))) != SKH_OK)  \
    {                                                \
        cleanup();                                   \
        return s;                                    \
    }
    LB_ALLOC(gbU, sizeof(uint32_t) * 6 * (size_t)nGroups);
    LB_ALLOC(keysA, sizeof(uint64_t) * (size_t)n);
    LB_ALLOC(keysB, sizeof(uint64_t) * (size_t)n);
    LB_ALLOC(valsB, sizeof(uint32_t) * (size_t)n);
    const uint32_t rsBlocks = (n + SKH_RS_THREADS * SKH_RS_ITEMS - 1) / (SKH_RS_THREADS * SKH_RS_ITEMS);
    LB_ALLOC(hist, sizeof(uint32_t) * 256 * (size_t)rsBlocks);
    // the digit histograms (256 x blocks words: 1.4 M for 23 M keys) are scanned by many workgroups -- blocks of 4096, a one-workgroup scan of
    // the block sums, an add pass -- not by ONE workgroup walking the whole array (round 3: up to 2.5 ms per sort pass, eight passes)
    const uint32_t histWords = 256u * rsBlocks, histBlocks = (histWords + SKH_SCAN_BLOCK * SKH_SCAN_ITEMS - 1) / (SKH_SCAN_BLOCK * SKH_SCAN_ITEMS);
    LB_ALLOC(histSums, sizeof(uint32_t) * (size_t)(histBlocks + 1));
    const uint32_t B = 256, G1 = (n + B - 1) / B;
    k_init_group_bounds<<<(nGroups * 6 + B - 1) / B, B, 0, st>>>(gbU.as<uint32_t>(), nGroups);
    k_group_bounds<<<(G1 + SKH_GB_RUN - 1) / SKH_GB_RUN, B, 0, st>>>(dBoxLo, dBoxHi, dGrp, n, gbU.as<uint32_t>()); // (a wave walks SKH_GB_RUN rows of 64)
    k_decode_group_bounds<<<(nGroups * 6 + B - 1) / B, B, 0, st>>>(gbU.as<uint32_t>(), out.groupBounds.as<float>(), nGroups);
    uint32_t* valsA = out.sortedVals.as<uint32_t>();
    // sort key: group id above a Morton code of `morton_bits` (10) bits per axis -- fewer when the group ids need the room
    uint32_t gbits = 0;
    while ((1ull << gbits) < (unsigned long long)nGroups)
        ++gbits;
    const uint32_t mb = std::min<uint32_t>(c->mortonBits, (64u - gbits) / 3u), keyShift = 3u * mb;
    k_morton<<<G1, B, 0, st>>>(dBoxLo, dBoxHi, dGrp, out.groupBounds.as<float>(), n, mb, keysA.as<uint64_t>(), valsA);
    // radix sort over the Morton bits + the group bits
    std::vector<uint32_t> shifts;
    for (uint32_t b = 0; b < keyShift + gbits; b += 8)
        shifts.push_back(b);
    uint64_t* kin = keysA.as<uint64_t>();
    uint64_t* kout = keysB.as<uint64_t>();
    uint32_t* vin = valsA;
    uint32_t* vout = valsB.as<uint32_t>();
    for (uint32_t shift : shifts)
    {
        k_rs_hist<<<rsBlocks, SKH_RS_THREADS, 0, st>>>(kin, n, shift, hist.as<uint32_t>(), rsBlocks, nullptr);
        if (histBlocks > 1)
        {
            k_scan_block<<<histBlocks, SKH_SCAN_BLOCK, 0, st>>>(hist.as<uint32_t>(), histWords, histSums.as<uint32_t>());
            k_rs_scan<<<1, 1024, 0, st>>>(histSums.as<uint32_t>(), histBlocks);
            k_scan_add<<<histBlocks, SKH_SCAN_BLOCK, 0, st>>>(hist.as<uint32_t>(), histWords, histSums.as<uint32_t>());
        }
        else
            k_rs_scan<<<1, 1024, 0, st>>>(hist.as<uint32_t>(), histWords);
        k_rs_scatter<<<rsBlocks, SKH_RS_THREADS, 0, st>>>(kin, vin, kout, vout, n, shift, hist.as<uint32_t>(), rsBlocks, nullptr);
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    if (vin != valsA) // odd number of passes: bring the values home
        SKH_TRY(c, hipMemcpyAsync(valsA, vin, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    const uint64_t* sortedKeys = kin;
    if (n >= 2)
    {
        DevBuf nodeSize, leafOrder, vals2, q[2], ctr, cLo[2], cHi[2], nn, pflags, ppos, scanSums;
        auto cleanup2 = [&]() {
            for (DevBuf* b : { &nodeSize, &leafOrder, &vals2, &q[0], &q[1], &ctr, &cLo[0], &cLo[1], &cHi[0], &cHi[1], &nn, &pflags, &ppos, &scanSums })
                dev_free(*b);
        };
#define LB2(buf, bytes)                              \
    if ((s = dev_alloc(c, buf, (bytes))) != SKH_OK)  \
    {                                                \
        cleanup2();                                  \
        cleanup();                                   \
        return s;                                    \
    }
        LB_ALLOC(childL, sizeof(int) * (size_t)n);
        LB_ALLOC(childR, sizeof(int) * (size_t)n);
        LB_ALLOC(nodeLo, sizeof(float4) * 2 * (size_t)n);
        LB_ALLOC(nodeHi, sizeof(float4) * 2 * (size_t)n);
        LB_ALLOC(gFirst, sizeof(uint32_t) * (size_t)nGroups);
        LB_ALLOC(gCount, sizeof(uint32_t) * (size_t)nGroups);
        LB2(nodeSize, sizeof(int) * (size_t)n);
        LB2(leafOrder, sizeof(uint32_t) * (size_t)n);
        LB2(vals2, sizeof(uint32_t) * (size_t)n);
        LB2(q[0], sizeof(CollapseItem) * (size_t)n);
        LB2(q[1], sizeof(CollapseItem) * (size_t)n);
        LB2(ctr, sizeof(uint32_t) * 4);
        SKH_TRY(c, hipMemcpyAsync(gFirst.p, groupFirst.data(), sizeof(uint32_t) * nGroups, hipMemcpyHostToDevice, st));
        SKH_TRY(c, hipMemcpyAsync(gCount.p, groupCount.data(), sizeof(uint32_t) * nGroups, hipMemcpyHostToDevice, st));
        SKH_TRY(c, hipMemcpyAsync(out.groupRoot.p, out.hostGroupRoot.data(), sizeof(int) * nGroups, hipMemcpyHostToDevice, st));
        const std::vector<int> preset = out.hostGroupRoot; // leaf refs of groups with <= leafMax primitives, INVALID otherwise
        hipError_t he = hipSuccess;
        if (ploc)
        {
            // ---- agglomerative clustering over the Morton order ----
            for (int k = 0; k < 2; ++k)
            {
                LB2(cLo[k], sizeof(float4) * (size_t)n);
                LB2(cHi[k], sizeof(float4) * (size_t)n);
            }
            LB2(nn, sizeof(uint32_t) * (size_t)n);
            LB2(pflags, sizeof(uint32_t) * (size_t)n);
            LB2(ppos, sizeof(uint32_t) * (size_t)n);
            LB2(scanSums, sizeof(uint32_t) * (size_t)(n / (SKH_SCAN_BLOCK * SKH_SCAN_ITEMS) + 2));
            uint32_t nonEmpty = 0;
            for (uint32_t g = 0; g < nGroups; ++g)
                nonEmpty += groupCount[g] ? 1u : 0u;
            uint32_t hc[4] = { 0, 0, 0, 0 }; // [2] PLOC node counter, [3] surviving cluster count
            he = hipMemcpyAsync(ctr.p, hc, sizeof(hc), hipMemcpyHostToDevice, st);
            if (riRounds)
            {
                LB_ALLOC(parent, sizeof(int) * 2 * (size_t)n);
                SKH_TRY(c, hipMemsetAsync(parent.p, 0xff, sizeof(int) * 2 * (size_t)n, st)); // -1: roots (and the ids PLOC never hands out) have no parent
            }
            k_ploc_init<<<G1, B, 0, st>>>(valsA, sortedKeys, dBoxLo, dBoxHi, n, cLo[0].as<float4>(), cHi[0].as<float4>(),
                                          nodeLo.as<float4>(), nodeHi.as<float4>(), keyShift);
            uint32_t m = n;
            int cur = 0;
            for (int iter = 0; he == hipSuccess && m > nonEmpty && iter < 4096; ++iter)
            {
                const uint32_t gm = (m + SKH_PLOC_BLOCK - 1) / SKH_PLOC_BLOCK;
                if (search == 1)
                    k_ploc_nn<SKH_PLOC_RADIUS_TLAS><<<gm, SKH_PLOC_BLOCK, 0, st>>>(cLo[cur].as<float4>(), cHi[cur].as<float4>(), m, nn.as<uint32_t>());
                else if (search == 2)
                    k_ploc_nn<SKH_PLOC_RADIUS_SEGS><<<gm, SKH_PLOC_BLOCK, 0, st>>>(cLo[cur].as<float4>(), cHi[cur].as<float4>(), m, nn.as<uint32_t>());
                else if (m <= c->plocTop)
                    // the top of the tree -- the levels every ray walks -- clustered with the wide search of the TLAS builder (few clusters left: cheap)
                    k_ploc_nn<SKH_PLOC_RADIUS_TLAS><<<gm, SKH_PLOC_BLOCK, 0, st>>>(cLo[cur].as<float4>(), cHi[cur].as<float4>(), m, nn.as<uint32_t>());
                else
                    k_ploc_nn<SKH_PLOC_RADIUS><<<gm, SKH_PLOC_BLOCK, 0, st>>>(cLo[cur].as<float4>(), cHi[cur].as<float4>(), m, nn.as<uint32_t>());
                k_ploc_merge<<<(m + B - 1) / B, B, 0, st>>>(cLo[cur].as<float4>(), cHi[cur].as<float4>(), nn.as<uint32_t>(), m, (int)n,
                                                            childL.as<int>(), childR.as<int>(), nodeSize.as<int>(), nodeLo.as<float4>(),
                                                            nodeHi.as<float4>(), ctr.as<uint32_t>() + 2, pflags.as<uint32_t>(), riRounds ? parent.as<int>() : nullptr);
                he = hipMemcpyAsync(ppos.p, pflags.p, sizeof(uint32_t) * (size_t)m, hipMemcpyDeviceToDevice, st);
                {
                    const uint32_t sb = (m + SKH_SCAN_BLOCK * SKH_SCAN_ITEMS - 1) / (SKH_SCAN_BLOCK * SKH_SCAN_ITEMS);
                    k_scan_block<<<sb, SKH_SCAN_BLOCK, 0, st>>>(ppos.as<uint32_t>(), m, scanSums.as<uint32_t>());
                    k_rs_scan<<<1, 1024, 0, st>>>(scanSums.as<uint32_t>(), sb);
                    k_scan_add<<<sb, SKH_SCAN_BLOCK, 0, st>>>(ppos.as<uint32_t>(), m, scanSums.as<uint32_t>());
                }
                k_ploc_compact<<<(m + B - 1) / B, B, 0, st>>>(cLo[cur].as<float4>(), cHi[cur].as<float4>(), pflags.as<uint32_t>(),
                                                              ppos.as<uint32_t>(), m, cLo[cur ^ 1].as<float4>(), cHi[cur ^ 1].as<float4>(),
                                                              ctr.as<uint32_t>() + 3);
                if (he == hipSuccess)
                    he = hipMemcpyAsync(hc, ctr.p, sizeof(hc), hipMemcpyDeviceToHost, st);
                if (he == hipSuccess)
                    he = hipStreamSynchronize(st);
                if (hc[3] >= m) // no progress: cannot happen (a mutually-nearest pair always exists), but never spin
                    break;
                m = hc[3];
                cur ^= 1;
            }
            if (he == hipSuccess)
                k_ploc_roots<<<(m + B - 1) / B, B, 0, st>>>(cLo[cur].as<float4>(), cHi[cur].as<float4>(), m, out.groupRoot.as<int>());
            if (he == hipSuccess && riRounds && n >= 8)
            {
                // ---- parallel reinsertion over the binary tree (skh_bvh.h): rounds of search / claim / own / check / apply / refit ----
                DevBuf moves, lock, moving, win, rflags, cost, cand, stamp, refitList, refitCount, activeA, activeB;
                auto cleanupR = [&]() {
                    for (DevBuf* b : { &moves, &lock, &moving, &win, &rflags, &cost, &cand, &stamp, &refitList, &refitCount, &activeA, &activeB })
                        dev_free(*b);
                };
                const size_t N2 = 2 * (size_t)n;
                skh_status sr = SKH_OK;
                if ((sr = dev_alloc(c, moves, sizeof(int4) * N2)) != SKH_OK || (sr = dev_alloc(c, lock, sizeof(unsigned long long) * N2)) != SKH_OK ||
                    (sr = dev_alloc(c, moving, sizeof(unsigned long long) * N2)) != SKH_OK || (sr = dev_alloc(c, win, N2)) != SKH_OK ||
                    (sr = dev_alloc(c, rflags, sizeof(uint32_t) * (size_t)n)) != SKH_OK || (sr = dev_alloc(c, cost, sizeof(double) * 2 + sizeof(uint32_t) * 2)) != SKH_OK ||
                    (sr = dev_alloc(c, cand, sizeof(uint32_t) * N2)) != SKH_OK || (sr = dev_alloc(c, stamp, sizeof(uint32_t) * (size_t)n)) != SKH_OK ||
                    (sr = dev_alloc(c, refitList, sizeof(uint32_t) * 2 * (size_t)n)) != SKH_OK || (sr = dev_alloc(c, refitCount, sizeof(uint32_t) * SKH_RI_MAX_LEVELS)) != SKH_OK ||
                    (sr = dev_alloc(c, activeA, N2)) != SKH_OK || (sr = dev_alloc(c, activeB, N2)) != SKH_OK)
                {
                    cleanupR();
                    cleanup2();
                    cleanup();
                    return sr;
                }
                const auto t0 = std::chrono::steady_clock::now();
                const uint32_t nInternal = hc[2]; // ids PLOC handed out
                const uint32_t GN = (uint32_t)((N2 + B - 1) / B);
                double* dCost = cost.as<double>();
                uint32_t* dWin = reinterpret_cast<uint32_t*>(dCost + 2); // [0] moves applied this round, [1] candidates
                he = hipMemsetAsync(cost.p, 0, sizeof(double) * 2 + sizeof(uint32_t) * 2, st);
                if (he == hipSuccess)
                    he = hipMemsetAsync(stamp.p, 0, sizeof(uint32_t) * (size_t)n, st);
                k_ri_cost<<<1024, B, 0, st>>>(nodeLo.as<float4>(), nodeHi.as<float4>(), (int)nInternal, dCost);
                out.riMinSize = riMinSize;
                uint32_t nCandHost = (uint32_t)N2; // (upper bound for the first round's list kernels; afterwards the previous round's count, which only shrinks... not relied on: see below)
                for (uint32_t r = 0; he == hipSuccess && r < riRounds; ++r)
                {
                    he = hipMemsetAsync(lock.p, 0, sizeof(unsigned long long) * N2, st);
                    if (he == hipSuccess)
                        he = hipMemsetAsync(moving.p, 0, sizeof(unsigned long long) * N2, st);
                    if (he == hipSuccess)
                        he = hipMemsetAsync(dWin, 0, sizeof(uint32_t) * 2, st);
                    // every third round searches from every node; the rounds between from last round's candidates and the nodes next to its moves only
                    uint8_t* activeCur = (r & 1u) ? activeB.as<uint8_t>() : activeA.as<uint8_t>();
                    uint8_t* activeNext = (r & 1u) ? activeA.as<uint8_t>() : activeB.as<uint8_t>();
                    if (he == hipSuccess)
                        he = hipMemsetAsync(activeNext, 0, N2, st);
                    k_ri_search<<<GN, B, 0, st>>>(childL.as<int>(), childR.as<int>(), parent.as<int>(), nodeSize.as<int>(), nodeLo.as<float4>(), nodeHi.as<float4>(), (int)n,
                                                  (int)riMinSize, (r % 3u) == 0u ? nullptr : activeCur, moves.as<int4>(), cand.as<uint32_t>(), dWin + 1);
                    // the list kernels are sized by the candidate count, read back here (one small synchronisation per round; the search is the long kernel)
                    if (he == hipSuccess)
                        he = hipMemcpyAsync(&nCandHost, dWin + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
                    if (he == hipSuccess)
                        he = hipStreamSynchronize(st);
                    out.riRounds = r + 1;
                    if (he != hipSuccess || nCandHost == 0)
                        break;
                    const uint32_t GC = (nCandHost + B - 1) / B;
                    k_ri_claim<<<GC, B, 0, st>>>(moves.as<int4>(), cand.as<uint32_t>(), dWin + 1, childL.as<int>(), childR.as<int>(), parent.as<int>(), lock.as<unsigned long long>(), activeNext);
                    k_ri_own<<<GC, B, 0, st>>>(moves.as<int4>(), cand.as<uint32_t>(), dWin + 1, childL.as<int>(), childR.as<int>(), parent.as<int>(), lock.as<unsigned long long>(),
                                               moving.as<unsigned long long>());
                    k_ri_check<<<GC, B, 0, st>>>(moves.as<int4>(), cand.as<uint32_t>(), dWin + 1, parent.as<int>(), moving.as<unsigned long long>(), win.as<uint8_t>(), dWin);
                    k_ri_apply<<<GC, B, 0, st>>>(moves.as<int4>(), cand.as<uint32_t>(), dWin + 1, win.as<uint8_t>(), childL.as<int>(), childR.as<int>(), parent.as<int>());
                    k_ri_mark<<<GC, B, 0, st>>>(moves.as<int4>(), cand.as<uint32_t>(), dWin + 1, win.as<uint8_t>(), parent.as<int>(), childL.as<int>(), childR.as<int>(), stamp.as<uint32_t>(), r + 1u, activeNext);
                    {
                        // refit of the stamped paths, one tree level per launch (rflags = stamped children still to come; lists ping-pong in `refitList`)
                        uint32_t* lv = refitCount.as<uint32_t>();
                        if (he == hipSuccess)
                            he = hipMemsetAsync(lv, 0, sizeof(uint32_t) * SKH_RI_MAX_LEVELS, st);
                        uint32_t* lists[2] = { refitList.as<uint32_t>(), refitList.as<uint32_t>() + n };
                        k_ri_pending<<<(n + B - 1) / B, B, 0, st>>>(childL.as<int>(), childR.as<int>(), stamp.as<uint32_t>(), r + 1u, (int)n, rflags.as<uint32_t>(), lists[0], lv);
                        // (the per-level counters are a RING: a binary tree deeper than SKH_RI_MAX_LEVELS -- nested "onion" geometry, chains the moves themselves
                        // grow -- wraps around it, each reused counter zeroed before its level runs, instead of failing the whole build for an optional
                        // pass: ADVICE r5.  A tree has at most n levels.)
                        bool levelsDone = false;
                        for (uint32_t k = 0; he == hipSuccess && k < n; ++k)
                        {
                            const uint32_t ci = k % SKH_RI_MAX_LEVELS, cn = (k + 1u) % SKH_RI_MAX_LEVELS;
                            if (k + 1u >= SKH_RI_MAX_LEVELS)
                                he = hipMemsetAsync(lv + cn, 0, sizeof(uint32_t), st);
                            k_ri_refit_level<<<256, B, 0, st>>>(lists[k & 1], lv + ci, lists[(k + 1) & 1], lv + cn, parent.as<int>(), childL.as<int>(), childR.as<int>(),
                                                                rflags.as<uint32_t>(), nodeLo.as<float4>(), nodeHi.as<float4>(), nodeSize.as<int>(), (int)n);
                            if ((k & 15u) == 15u)
                            {
                                uint32_t next = 0;
                                if (he == hipSuccess)
                                    he = hipMemcpyAsync(&next, lv + cn, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
                                if (he == hipSuccess)
                                    he = hipStreamSynchronize(st);
                                if (next == 0)
                                {
                                    levelsDone = true;
                                    break;
                                }
                            }
                        }
                        if (he == hipSuccess && !levelsDone)
                        {
                            cleanupR();
                            cleanup2();
                            cleanup();
                            c->err = "lbvh_build: the reinsertion pass's refit did not finish within n levels (a cycle in the tree?)";
                            return SKH_FAIL;
                        }
                    }
                    uint32_t moved = 0;
                    if (he == hipSuccess)
                        he = hipMemcpyAsync(&moved, dWin, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
                    if (he == hipSuccess)
                        he = hipStreamSynchronize(st);
                    out.riMoves += moved;
                    if (moved == 0)
                        break;
                }
                if (he == hipSuccess)
                {
                    double hcost[2] = { 0, 0 };
                    k_ri_cost<<<1024, B, 0, st>>>(nodeLo.as<float4>(), nodeHi.as<float4>(), (int)nInternal, dCost + 1);
                    he = hipMemcpyAsync(hcost, dCost, sizeof(hcost), hipMemcpyDeviceToHost, st);
                    if (he == hipSuccess)
                        he = hipStreamSynchronize(st);
                    out.riCostBefore = hcost[0], out.riCostAfter = hcost[1];
                }
                out.riMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                cleanupR();
            }
        }
        else
        {
            // ---- Karras radix tree ----
            LB_ALLOC(parent, sizeof(int) * 2 * (size_t)n);
            LB_ALLOC(rangeF, sizeof(int) * (size_t)n);
            LB_ALLOC(rangeL, sizeof(int) * (size_t)n);
            LB_ALLOC(flags, sizeof(uint32_t) * (size_t)n);
            SKH_TRY(c, hipMemsetAsync(flags.p, 0, sizeof(uint32_t) * (size_t)n, st));
            k_karras<<<(n - 1 + B - 1) / B, B, 0, st>>>(sortedKeys, (int)n, childL.as<int>(), childR.as<int>(), parent.as<int>(),
                                                       rangeF.as<int>(), rangeL.as<int>());
            k_refit<<<G1, B, 0, st>>>(valsA, dBoxLo, dBoxHi, parent.as<int>(), childL.as<int>(), childR.as<int>(), flags.as<uint32_t>(),
                                      nodeLo.as<float4>(), nodeHi.as<float4>(), (int)n);
            k_sizes_from_ranges<<<(n - 1 + B - 1) / B, B, 0, st>>>(rangeF.as<int>(), rangeL.as<int>(), (int)n, nodeSize.as<int>());
            k_group_roots<<<(n - 1 + B - 1) / B, B, 0, st>>>(rangeF.as<int>(), rangeL.as<int>(), sortedKeys, gFirst.as<uint32_t>(),
                                                            gCount.as<uint32_t>(), (int)n, leafMax, keyShift, out.groupRoot.as<int>());
        }
        // group roots come back as BINARY node ids; they seed the level-by-level collapse into 4-wide nodes
        if (he != hipSuccess || hipMemcpyAsync(out.hostGroupRoot.data(), out.groupRoot.p, sizeof(int) * nGroups, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
        {
            cleanup2();
            cleanup();
            c->err = "lbvh_build: tree build / root read-back failed";
            return SKH_FAIL;
        }
        std::vector<CollapseItem> seed;
        for (uint32_t g = 0; g < nGroups; ++g)
        {
            if (groupCount[g] <= (uint32_t)leafMax)
                out.hostGroupRoot[g] = preset[g]; // whole group is one leaf (or empty)
            else
            {
                seed.push_back(CollapseItem{ out.hostGroupRoot[g], (int)seed.size(), (int)groupFirst[g], 0 });
                out.hostGroupRoot[g] = (int)seed.size() - 1;
            }
        }
        uint32_t cnt = (uint32_t)seed.size();
        uint32_t hctr[2] = { cnt, 0 }; // [0] next free output slot, [1] next-level queue length
        if (cnt)
            he = hipMemcpyAsync(q[0].p, seed.data(), sizeof(CollapseItem) * cnt, hipMemcpyHostToDevice, st);
        if (he == hipSuccess)
            he = hipMemcpyAsync(ctr.p, hctr, sizeof(hctr), hipMemcpyHostToDevice, st);
        if (he == hipSuccess)
            he = hipMemcpyAsync(out.groupRoot.p, out.hostGroupRoot.data(), sizeof(int) * nGroups, hipMemcpyHostToDevice, st);
        k_iota<<<G1, B, 0, st>>>(leafOrder.as<uint32_t>(), n);
        int cur = 0;
        out.levelStart.assign(1, 0u);
        out.levelStart.push_back(cnt); // (level 0 = the groups' roots)
        while (he == hipSuccess && cnt > 0)
        {
            k_collapse<<<(cnt + B - 1) / B, B, 0, st>>>(q[cur].as<CollapseItem>(), cnt, childL.as<int>(), childR.as<int>(), nodeSize.as<int>(),
                                                           nodeLo.as<float4>(), nodeHi.as<float4>(), (int)n, leafMax, out.nodes.p, ctr.as<uint32_t>(),
                                                           q[cur ^ 1].as<CollapseItem>(), ctr.as<uint32_t>() + 1, leafOrder.as<uint32_t>(), search == 0 ? c->splitPairs * 0.1f : 0.0f);
            he = hipMemcpyAsync(hctr, ctr.p, sizeof(hctr), hipMemcpyDeviceToHost, st);
            if (he == hipSuccess)
                he = hipStreamSynchronize(st);
            cnt = hctr[1];
            if (cnt)
                out.levelStart.push_back(hctr[0]); // (the next level's nodes took the slots up to here)
            const uint32_t zero = 0;
            if (he == hipSuccess)
                he = hipMemcpyAsync(ctr.as<uint32_t>() + 1, &zero, sizeof(uint32_t), hipMemcpyHostToDevice, st);
            if (he == hipSuccess)
                he = hipStreamSynchronize(st);
            cur ^= 1;
        }
        if (he == hipSuccess)
        {
            // primitives into final leaf order
            k_permute_u32<<<G1, B, 0, st>>>(valsA, leafOrder.as<uint32_t>(), n, vals2.as<uint32_t>());
            he = hipMemcpyAsync(valsA, vals2.p, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st);
            if (he == hipSuccess)
                he = hipStreamSynchronize(st);
        }
        if (he == hipSuccess && lineRecBytes)
        {
            // ---- leaves by 128-byte line (skh_bvh.h: k_leaf_place): padding slots in front of the leaves that would straddle ----
            DevBuf leafCnt, chunk, remap, sums, vals3;
            auto cleanup3 = [&]() {
                for (DevBuf* b : { &leafCnt, &chunk, &remap, &sums, &vals3 })
                    dev_free(*b);
            };
            const uint32_t nCh = (n + SKH_LEAF_CHUNK - 1u) / SKH_LEAF_CHUNK;
            const uint32_t sb = (nCh + SKH_SCAN_BLOCK * SKH_SCAN_ITEMS - 1) / (SKH_SCAN_BLOCK * SKH_SCAN_ITEMS);
            skh_status s3 = SKH_OK;
            if ((s3 = dev_alloc(c, leafCnt, (size_t)n)) != SKH_OK || (s3 = dev_alloc(c, chunk, sizeof(uint32_t) * (size_t)nCh)) != SKH_OK ||
                (s3 = dev_alloc(c, remap, sizeof(uint32_t) * (size_t)n)) != SKH_OK || (s3 = dev_alloc(c, sums, sizeof(uint32_t) * ((size_t)sb + 2))) != SKH_OK)
            {
                cleanup3();
                cleanup2();
                cleanup();
                return s3;
            }
            int* nodeRefs = reinterpret_cast<int*>(out.nodes.p) + 12; // Node4::child
            const uint32_t nRefs = hctr[0] * 4u;
            he = hipMemsetAsync(leafCnt.p, 0, (size_t)n, st);
            if (nRefs)
                k_leaf_mark<<<(nRefs + B - 1) / B, B, 0, st>>>(nodeRefs, nRefs, 16u, 4u, leafCnt.as<uint8_t>());
            k_leaf_mark<<<(nGroups + B - 1) / B, B, 0, st>>>(out.groupRoot.as<int>(), nGroups, 1u, 1u, leafCnt.as<uint8_t>());
            k_leaf_place<<<(nCh + B - 1) / B, B, 0, st>>>(leafCnt.as<uint8_t>(), n, lineRecBytes, nullptr, chunk.as<uint32_t>(), nullptr);
            uint32_t lastLen = 0, lastBase = 0;
            if (he == hipSuccess)
                he = hipMemcpyAsync(&lastLen, chunk.as<uint32_t>() + (nCh - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, st);
            k_scan_block<<<sb, SKH_SCAN_BLOCK, 0, st>>>(chunk.as<uint32_t>(), nCh, sums.as<uint32_t>());
            k_rs_scan<<<1, 1024, 0, st>>>(sums.as<uint32_t>(), sb);
            k_scan_add<<<sb, SKH_SCAN_BLOCK, 0, st>>>(chunk.as<uint32_t>(), nCh, sums.as<uint32_t>());
            if (he == hipSuccess)
                he = hipMemcpyAsync(&lastBase, chunk.as<uint32_t>() + (nCh - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, st);
            if (he == hipSuccess)
                he = hipStreamSynchronize(st);
            const uint64_t total = (uint64_t)lastBase + lastLen;
            if (he == hipSuccess && total >= (1ull << 28))
            {
                cleanup3();
                cleanup2();
                cleanup();
                c->err = "lbvh_build: more than 2^28 leaf slots";
                return SKH_INVALID_ARGUMENT;
            }
            if (he == hipSuccess && (s3 = dev_alloc(c, vals3, sizeof(uint32_t) * (size_t)std::max<uint64_t>(1, total))) != SKH_OK)
            {
                cleanup3();
                cleanup2();
                cleanup();
                return s3;
            }
            if (he == hipSuccess)
            {
                k_leaf_place<<<(nCh + B - 1) / B, B, 0, st>>>(leafCnt.as<uint8_t>(), n, lineRecBytes, chunk.as<uint32_t>(), nullptr, remap.as<uint32_t>());
                if (nRefs)
                    k_leaf_patch<<<(nRefs + B - 1) / B, B, 0, st>>>(nodeRefs, nRefs, 16u, 4u, remap.as<uint32_t>());
                k_leaf_patch<<<(nGroups + B - 1) / B, B, 0, st>>>(out.groupRoot.as<int>(), nGroups, 1u, 1u, remap.as<uint32_t>());
                he = hipMemsetAsync(vals3.p, 0xff, sizeof(uint32_t) * (size_t)total, st);
                k_leaf_scatter<<<G1, B, 0, st>>>(valsA, remap.as<uint32_t>(), n, vals3.as<uint32_t>());
                if (he == hipSuccess)
                    he = hipMemcpyAsync(out.hostGroupRoot.data(), out.groupRoot.p, sizeof(int) * nGroups, hipMemcpyDeviceToHost, st);
                if (he == hipSuccess)
                    he = hipStreamSynchronize(st);
                if (he == hipSuccess)
                {
                    std::swap(out.sortedVals, vals3); // (the unpadded order is freed with vals3)
                    out.numSlots = (uint32_t)total;
                }
            }
            cleanup3();
        }
        cleanup2();
        if (he != hipSuccess)
        {
            cleanup();
            c->err = std::string("lbvh_build collapse: ") + hipGetErrorString(he);
            return SKH_FAIL;
        }
        out.numNodes = hctr[0];
#undef LB2
    }
    else
        SKH_TRY(c, hipMemcpyAsync(out.groupRoot.p, out.hostGroupRoot.data(), sizeof(int) * nGroups, hipMemcpyHostToDevice, st));
    hipError_t e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess)
    {
        c->err = std::string("lbvh_build: ") + hipGetErrorString(e);
        return SKH_FAIL;
    }
    e = hipGetLastError();
    if (e != hipSuccess)
    {
        c->err = std::string("lbvh_build launch: ") + hipGetErrorString(e);
        return SKH_FAIL;
    }
    return SKH_OK;
#undef LB_ALLOC
}

// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// TLAS by full-sweep SAH on the host.  The top level holds a few thousand instance boxes (the reference's IAS,
// OptixRender.cpp:443-495) and every ray walks it, so its quality matters far more than its build time; an exact
// sweep over three axes costs O(n log^2 n) on a few thousand boxes -- well under a millisecond per thousand.
// Invalid (masked-out) instances are left out.  Output: 64-byte nodes + leaf order, same encoding as the GPU LBVH.
// ---------------------------------------------------------------------------------------------------------------
struct HostBox
{
    float lo[3], hi[3];
};
static inline float hb_half_area(const float* lo, const float* hi)
{
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return ex * ey + ey * ez + ez * ex;
}
struct HostBin // internal node of the host-side binary tree
{
    int left, right; // >= 0 HostBin index, < 0 leaf ref
    float lo[3], hi[3];
};
static void hb_box_of_ref(int ref, const std::vector<HostBin>& bin, const std::vector<HostBox>& boxes, float* lo, float* hi)
{
    if (ref >= 0)
    {
        memcpy(lo, bin[ref].lo, sizeof(float) * 3);
        memcpy(hi, bin[ref].hi, sizeof(float) * 3);
    }
    else
    {
        const HostBox& b = boxes[((uint32_t)~ref) >> 3];
        memcpy(lo, b.lo, sizeof(float) * 3);
        memcpy(hi, b.hi, sizeof(float) * 3);
    }
}
static int tlas_sah_build(const std::vector<HostBox>& boxes, const std::vector<uint32_t>& ids, std::vector<Node4>& nodes,
                          std::vector<uint32_t>& order)
{
    const uint32_t n = (uint32_t)ids.size();
    nodes.clear();
    order.clear();
    if (n == 0)
        return SKH_REF_INVALID;
    order.assign(ids.begin(), ids.end());
    if (n == 1)
        return ~(int)((ids[0] << 3) | 0u);
    std::vector<HostBin> bin;
    struct Task
    {
        uint32_t first, count;
        int parent;
        int side;
    };
    std::vector<Task> stack;
    stack.push_back(Task{ 0, n, -1, 0 });
    std::vector<float> rightArea;
    int rootRef = 0;
    while (!stack.empty())
    {
        const Task t = stack.back();
        stack.pop_back();
        int ref;
        if (t.count == 1)
            ref = ~(int)((order[t.first] << 3) | 0u); // leaf ref carries the INSTANCE ID (one instance per leaf)
        else
        {
            float bestCost = INFINITY;
            int bestAxis = 0;
            uint32_t bestSplit = t.count / 2;
            auto sortAxis = [&](int axis) {
                std::sort(order.begin() + t.first, order.begin() + t.first + t.count, [&](uint32_t a, uint32_t b) {
                    const float ca = boxes[a].lo[axis] + boxes[a].hi[axis], cb = boxes[b].lo[axis] + boxes[b].hi[axis];
                    return ca < cb || (ca == cb && a < b);
                });
            };
            for (int axis = 0; axis < 3; ++axis)
            {
                sortAxis(axis);
                rightArea.assign(t.count + 1, 0.0f);
                float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
                for (uint32_t i = t.count; i-- > 1;)
                {
                    const HostBox& b = boxes[order[t.first + i]];
                    for (int k = 0; k < 3; ++k)
                    {
                        lo[k] = std::min(lo[k], b.lo[k]);
                        hi[k] = std::max(hi[k], b.hi[k]);
                    }
                    rightArea[i] = hb_half_area(lo, hi);
                }
                for (int k = 0; k < 3; ++k)
                {
                    lo[k] = INFINITY;
                    hi[k] = -INFINITY;
                }
                for (uint32_t i = 1; i < t.count; ++i)
                {
                    const HostBox& b = boxes[order[t.first + i - 1]];
                    for (int k = 0; k < 3; ++k)
                    {
                        lo[k] = std::min(lo[k], b.lo[k]);
                        hi[k] = std::max(hi[k], b.hi[k]);
                    }
                    const float cost = hb_half_area(lo, hi) * (float)i + rightArea[i] * (float)(t.count - i);
                    if (cost < bestCost)
                    {
                        bestCost = cost;
                        bestAxis = axis;
                        bestSplit = i;
                    }
                }
            }
            sortAxis(bestAxis);
            ref = (int)bin.size();
            HostBin hbn;
            hbn.left = hbn.right = SKH_REF_INVALID;
            for (int k = 0; k < 3; ++k)
            {
                hbn.lo[k] = INFINITY;
                hbn.hi[k] = -INFINITY;
            }
            for (uint32_t i = t.first; i < t.first + t.count; ++i)
                for (int k = 0; k < 3; ++k)
                {
                    hbn.lo[k] = std::min(hbn.lo[k], boxes[order[i]].lo[k]);
                    hbn.hi[k] = std::max(hbn.hi[k], boxes[order[i]].hi[k]);
                }
            bin.push_back(hbn);
            stack.push_back(Task{ t.first + bestSplit, t.count - bestSplit, ref, 1 });
            stack.push_back(Task{ t.first, bestSplit, ref, 0 });
        }
        if (t.parent < 0)
            rootRef = ref;
        else if (t.side == 0)
            bin[t.parent].left = ref;
        else
            bin[t.parent].right = ref;
    }
    // collapse the binary tree into 4-wide nodes (same greedy rule as k_collapse)
    struct Item
    {
        int bin, out;
    };
    std::vector<Item> work;
    nodes.push_back(Node4{});
    work.push_back(Item{ rootRef, 0 });
    for (size_t w = 0; w < work.size(); ++w)
    {
        const Item it = work[w];
        int slot[4];
        int cnt = 2;
        slot[0] = bin[it.bin].left;
        slot[1] = bin[it.bin].right;
        while (cnt < 4)
        {
            int best = -1;
            float bestA = -1.0f;
            for (int k = 0; k < cnt; ++k)
                if (slot[k] >= 0)
                {
                    const float a = hb_half_area(bin[slot[k]].lo, bin[slot[k]].hi);
                    if (a > bestA)
                    {
                        bestA = a;
                        best = k;
                    }
                }
            if (best < 0)
                break;
            const int cidx = slot[best];
            slot[best] = bin[cidx].left;
            slot[cnt++] = bin[cidx].right;
        }
        float clo[4][3], chi[4][3];
        int refs[4];
        for (int k = 0; k < cnt; ++k)
        {
            hb_box_of_ref(slot[k], bin, boxes, clo[k], chi[k]);
            if (slot[k] >= 0)
            {
                refs[k] = (int)nodes.size();
                nodes.push_back(Node4{});
                work.push_back(Item{ slot[k], refs[k] });
            }
            else
                refs[k] = slot[k];
        }
        Node4 nd;
        encode_node4(nd, bin[it.bin].lo, bin[it.bin].hi, clo, chi, refs, cnt);
        nodes[it.out] = nd;
    }
    return 0; // the root is node 0
}

// The entry points below take C linkage from their declarations in include/strelka_hip.h.

// Ends the speculation.  Whatever was traced ahead and not yet delivered is dropped: its share of the pass's rays (sub-frames of a
// pass trace the same pixels with different samples: equal shares to a fraction of a per cent) leaves the ray counts.
static void spec_drop(skh_context* c, bool keepLast = false)
{
    skh_context::Speculation& sp = c->spec;
    if (sp.nextInFlight)
    {
        // a whole pass traced ahead that nobody will collect: wait for it (its buffers are about to be reused), take its rays out again
        unsigned long long after[2] = { 0, 0 };
        (void)hipStreamSynchronize(c->stream);
        if (hipMemcpy(after, c->dStats.p, sizeof(after), hipMemcpyDeviceToHost) == hipSuccess)
        {
            c->discardedRadiance += after[0] - sp.statsMark[0];
            c->discardedShadow += after[1] - sp.statsMark[1];
        }
        c->discardedSubframes += sp.nextCount;
        sp.nextInFlight = false;
    }
    if (sp.valid && sp.consumed < sp.count && sp.count > 0)
    {
        const uint32_t left = sp.count - sp.consumed;
        c->discardedRadiance += sp.passRadiance * left / sp.count;
        c->discardedShadow += sp.passShadow * left / sp.count;
        c->discardedSubframes += left;
    }
    sp.valid = false;
    sp.buf = 0;
    if (!keepLast)
        sp.haveLast = false;
}

uint32_t skh_abi_version(void)
{
    return SKH_ABI_VERSION;
}

skh_status skh_create(int device_ordinal, skh_context** out_ctx)
{
    if (!out_ctx)
        return SKH_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_ordinal < 0 || device_ordinal >= count)
        return SKH_FAIL; // no GPU: the product path fails loudly, there is no CPU fallback
    skh_context* c = new skh_context();
    c->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess || hipStreamCreate(&c->stream2) != hipSuccess || hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->evShade, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->evShadow, hipEventDisableTiming) != hipSuccess)
    {
        delete c;
        return SKH_FAIL;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess)
        c->numCUs = prop.multiProcessorCount;
    uint32_t tab[5][32];
    init_sobol_table(tab);
    std::vector<uint32_t> lut(SKH_SOBOL_LUT_WORDS);
    for (uint32_t d = 0; d < 5; ++d)
        for (uint32_t b = 0; b < 4; ++b)
            for (uint32_t v = 0; v < 256; ++v)
            {
                uint32_t x = 0;
                for (uint32_t j = 0; j < 8; ++j)
                    if ((v >> j) & 1u)
                        x ^= tab[d][8 * b + j];
                lut[(d * 4 + b) * 256 + v] = x;
            }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_sobol), tab, sizeof(tab)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_sobol_lut), lut.data(), lut.size() * sizeof(uint32_t)) != hipSuccess)
    {
        delete c;
        return SKH_FAIL;
    }
    if (dev_alloc(c, c->dStats, sizeof(StatsDev)) != SKH_OK || hipMemset(c->dStats.p, 0, sizeof(StatsDev)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->hOverflow), 64, hipHostMallocMapped) != hipSuccess)
    {
        dev_free(c->dStats);
        delete c;
        return SKH_FAIL;
    }
    *c->hOverflow = 0u;
    *out_ctx = c;
    return SKH_OK;
}

void skh_destroy(skh_context* c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream3)
        (void)hipStreamSynchronize(c->stream3);
    if (c->comm)
        (void)skh_comm_destroy(c);
    for (DevBuf* b : { &c->dShadeTris, &c->dShadeInst, &c->dVerts, &c->dIndices, &c->dMeshes, &c->dPoints, &c->dRadii, &c->dInstances, &c->dLights, &c->dMaterials, &c->dHairConst,
                       &c->dCurveSegBase, &c->dSegStartAll, &c->dTriNodes, &c->dTris, &c->dSegNodes, &c->dSegs,
                       &c->dTlasNodes, &c->dTlasInst, &c->dDevInst, &c->dTravInst, &c->dTexels, &c->dTexDesc, &c->dTriOrder, &c->dTriMeshK, &c->dTriLocalK, &c->dWInstK, &c->dWFirstK, &c->dTriNodeBox, &c->dSegOrder, &c->dSegBuildStartK, &c->dSegLocalK, &c->dSegInstOfK, &c->dSegNodeBox, &c->dScatterXY, &c->dRaygenBase, &c->dTileXY, &c->dAccum, &c->dDiffuse, &c->dSpecular, &c->dDiffCnt,
                       &c->dSpecCnt, &c->dSums, &c->dPath, &c->dRayQ[0], &c->dRayQ[1], &c->dHits, &c->dShadowQ, &c->dContrib,
                       &c->dCounts, &c->dOvf, &c->dOvf2, &c->dStats, &c->dScratchImage, &c->dPathB })
        dev_free(*b);
    for (hipEvent_t e : c->eventPool)
        (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    if (c->stream2)
        (void)hipStreamDestroy(c->stream2);
    if (c->stream3)
        (void)hipStreamDestroy(c->stream3);
    if (c->evShade)
        (void)hipEventDestroy(c->evShade);
    if (c->evShadow)
        (void)hipEventDestroy(c->evShadow);
    if (c->hOverflow)
        (void)hipHostFree(c->hOverflow);
    dev_free(c->dTileSend);
    delete c;
}

const char* skh_last_error(const skh_context* c)
{
    return c ? c->err.c_str() : "null context";
}

skh_status skh_set_geometry(skh_context* c, const skh_vertex* verts, uint32_t n_verts, const uint32_t* indices, uint32_t n_indices,
                            const skh_mesh* meshes, uint32_t n_meshes)
{
    if (!c || (n_verts && !verts) || (n_indices && !indices) || (n_meshes && !meshes))
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    (void)hipSetDevice(c->device);
    for (uint32_t m = 0; m < n_meshes; ++m)
    {
        const skh_mesh& me = meshes[m];
        if ((uint64_t)me.index_offset + me.index_count > n_indices || (uint64_t)me.vertex_offset + me.vertex_count > n_verts ||
            me.index_count % 3 != 0)
        {
            c->err = "skh_set_geometry: mesh " + std::to_string(m) + " is out of range";
            return SKH_INVALID_ARGUMENT;
        }
        // indices are mesh-local (closest_hit.cu:365-376 adds mVbOffset): every one must address a vertex of its own mesh
        const uint32_t* ib = indices + me.index_offset;
        uint32_t worst = 0;
        for (uint32_t k = 0; k < me.index_count; ++k)
            worst = std::max(worst, ib[k]);
        if (me.index_count && worst >= me.vertex_count)
        {
            c->err = "skh_set_geometry: mesh " + std::to_string(m) + " has index " + std::to_string(worst) + " >= its vertex_count " +
                     std::to_string(me.vertex_count);
            return SKH_INVALID_ARGUMENT;
        }
    }
    c->meshes.assign(meshes, meshes + n_meshes);
    c->nVerts = n_verts;
    c->nIndices = n_indices;
    c->accelBuilt = false;
    {
        // signature of the TOPOLOGY (mesh table + index buffer, FNV-1a over their words): skh_refit_accel keeps the hierarchy only while it is the built one
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const uint32_t* w, size_t nw) {
            for (size_t k = 0; k < nw; ++k)
                h = (h ^ w[k]) * 1099511628211ull;
        };
        mix(reinterpret_cast<const uint32_t*>(meshes), sizeof(skh_mesh) / 4 * (size_t)n_meshes);
        mix(indices, n_indices);
        c->geomSig = h ^ ((uint64_t)n_verts << 32);
    }
    skh_status s;
    if ((s = dev_upload(c, c->dVerts, verts, sizeof(skh_vertex) * (size_t)n_verts)) != SKH_OK)
        return s;
    if ((s = dev_upload(c, c->dIndices, indices, sizeof(uint32_t) * (size_t)n_indices)) != SKH_OK)
        return s;
    return dev_upload(c, c->dMeshes, meshes, sizeof(skh_mesh) * (size_t)n_meshes);
}

skh_status skh_set_curves(skh_context* c, const float* points_xyz, uint32_t n_points, const float* radii, uint32_t n_radii,
                          const uint32_t* vertex_counts, uint32_t n_vertex_counts, const skh_curve* curves, uint32_t n_curves)
{
    if (!c || (n_points && !points_xyz) || (n_radii && !radii) || (n_vertex_counts && !vertex_counts) || (n_curves && !curves))
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    if (n_radii != n_points)
    {
        c->err = "skh_set_curves: one radius per control point is required";
        return SKH_INVALID_ARGUMENT;
    }
    (void)hipSetDevice(c->device);
    for (uint32_t k = 0; k < n_curves; ++k)
    {
        // ranges of oka::Curve (scene.h:29-42) against the arrays they address; the strands of a set must fit its point range
        const skh_curve& cu = curves[k];
        bool ok = (uint64_t)cu.vertex_counts_start + cu.vertex_counts_count <= n_vertex_counts &&
                  (uint64_t)cu.points_start + cu.points_count <= n_points && (uint64_t)cu.widths_start + cu.widths_count <= n_radii;
        uint64_t used = 0;
        for (uint32_t j = 0; ok && j < cu.vertex_counts_count; ++j)
            used += vertex_counts[cu.vertex_counts_start + j];
        if (!ok || used > cu.points_count)
        {
            c->err = "skh_set_curves: curve set " + std::to_string(k) + " addresses data outside the arrays passed in";
            return SKH_INVALID_ARGUMENT;
        }
    }
    c->curves.assign(curves, curves + n_curves);
    c->curveVertexCounts.assign(vertex_counts, vertex_counts + n_vertex_counts);
    c->nPoints = n_points;
    c->accelBuilt = false;
    {
        // signature of the curve sets' TOPOLOGY (set table + vertex counts + array sizes): control points and radii may change under skh_refit_accel
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const uint32_t* w, size_t nw) {
            for (size_t k = 0; k < nw; ++k)
                h = (h ^ w[k]) * 1099511628211ull;
        };
        mix(reinterpret_cast<const uint32_t*>(curves), sizeof(skh_curve) / 4 * (size_t)n_curves);
        mix(vertex_counts, n_vertex_counts);
        c->curveSig = h ^ ((uint64_t)n_points << 32);
        c->curvePointsEdited = true;
    }
    skh_status s;
    if ((s = dev_upload(c, c->dPoints, points_xyz, sizeof(float) * 3 * (size_t)n_points)) != SKH_OK)
        return s;
    return dev_upload(c, c->dRadii, radii, sizeof(float) * (size_t)n_radii);
}

skh_status skh_set_instances(skh_context* c, const skh_instance* instances, uint32_t n)
{
    if (!c || (n && !instances))
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    (void)hipSetDevice(c->device);
    // (the same table again -- a caller that re-sends the whole scene with edited vertices -- keeps a refit possible)
    if (!(n == c->nInstances && n == c->instances.size() && (n == 0 || memcmp(c->instances.data(), instances, sizeof(skh_instance) * (size_t)n) == 0)))
        c->refitReady = false;
    c->instances.assign(instances, instances + n);
    c->nInstances = n;
    c->accelBuilt = false;
    return dev_upload(c, c->dInstances, instances, sizeof(skh_instance) * (size_t)n);
}

skh_status skh_set_lights(skh_context* c, const skh_light* lights, uint32_t n)
{
    if (!c || (n && !lights))
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    (void)hipSetDevice(c->device);
    c->nLights = n;
    return dev_upload(c, c->dLights, lights, sizeof(skh_light) * (size_t)n);
}

skh_status skh_set_textures(skh_context* c, const skh_texture* textures, uint32_t n)
{
    if (!c || (n && !textures))
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    (void)hipSetDevice(c->device);
    std::vector<uint4> desc(n);
    uint64_t total = 0;
    for (uint32_t k = 0; k < n; ++k)
    {
        if (!textures[k].rgba8 || textures[k].width == 0 || textures[k].height == 0)
        {
            c->err = "skh_set_textures: texture " + std::to_string(k) + " is empty";
            return SKH_INVALID_ARGUMENT;
        }
        desc[k] = make_uint4((uint32_t)total, textures[k].width, textures[k].height, 0u);
        total += (uint64_t)textures[k].width * textures[k].height;
        if (total >= (1ull << 32))
        {
            c->err = "skh_set_textures: more than 2^32 texels";
            return SKH_INVALID_ARGUMENT;
        }
    }
    std::vector<uint32_t> texels((size_t)total);
    for (uint32_t k = 0; k < n; ++k)
        memcpy(texels.data() + desc[k].x, textures[k].rgba8, (size_t)textures[k].width * textures[k].height * 4);
    skh_status s = dev_upload(c, c->dTexels, texels.data(), sizeof(uint32_t) * texels.size());
    if (s == SKH_OK)
        s = dev_upload(c, c->dTexDesc, desc.data(), sizeof(uint4) * desc.size());
    c->nTextures = s == SKH_OK ? n : 0;
    return s;
}
skh_status skh_set_materials(skh_context* c, const skh_material* materials, uint32_t n)
{
    if (!c || (n && !materials))
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    (void)hipSetDevice(c->device);
    c->nMaterials = n;
    c->hasHairMaterial = false;
    for (uint32_t k = 0; k < n; ++k)
        c->hasHairMaterial = c->hasHairMaterial || materials[k].type == SKH_MAT_HAIR;
    skh_status s = dev_upload(c, c->dMaterials, materials, sizeof(skh_material) * (size_t)n);
    if (s != SKH_OK)
        return s;
    // the material-only terms of the hair BSDF, computed once per material ON THE DEVICE by the code the per-call path ran (skh_device.h hair_const)
    if ((s = dev_alloc(c, c->dHairConst, sizeof(HairConst) * (size_t)std::max(1u, n))) != SKH_OK)
        return s;
    if (n)
        k_hair_consts<<<(n + 63) / 64, 64, 0, c->stream>>>(c->dMaterials.as<Material>(), n, c->dHairConst.as<HairConst>());
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    return SKH_OK;
}

static skh_status build_shading_tables(skh_context* c)
{
    const uint32_t nMeshes = (uint32_t)c->meshes.size();
    std::vector<uint32_t> base(nMeshes + 1u, 0u);
    uint64_t total = 0; // (summed wide and tested before narrowing: a 32-bit sum that wraps would pass the test below with aliased bases)
    for (uint32_t m = 0; m < nMeshes; ++m)
    {
        total += c->meshes[m].index_count / 3u;
        base[m + 1] = (uint32_t)std::min<uint64_t>(total, 0xffffffffull);
    }
    const uint32_t nTris = base[nMeshes];
    c->nShadeRecords = nTris;
    skh_status s;
    if (total >= SKH_PRIM_DIRECT)
    {
        c->err = "skh_build_accel: more than 2^31 - 1 triangles (the primitive word of a hit keeps its top bit for SKH_PRIM_DIRECT)";
        return SKH_INVALID_ARGUMENT;
    }
    if ((s = dev_alloc(c, c->dShadeTris, std::max<size_t>(96, (size_t)nTris * 96))) != SKH_OK)
        return s;
    if (nTris)
    {
        DevBuf dBase;
        if ((s = dev_upload(c, dBase, base.data(), sizeof(uint32_t) * base.size())) != SKH_OK)
            return s;
        k_gather_shade_tris<<<(nTris + 255u) / 256u, 256, 0, c->stream>>>(c->dVerts.as<uint8_t>(), c->dIndices.as<uint32_t>(), c->dMeshes.as<uint4>(),
                                                                         dBase.as<uint32_t>(), nMeshes, nTris, c->dShadeTris.as<float4>());
        const hipError_t e = hipStreamSynchronize(c->stream);
        dev_free(dBase);
        if (e != hipSuccess)
        {
            c->err = std::string("k_gather_shade_tris: ") + hipGetErrorString(e);
            return SKH_FAIL;
        }
    }
    std::vector<skh_instance> inst = c->instances;
    for (skh_instance& i : inst)
        if (i.type == SKH_INSTANCE_MESH)
            i.light_id = i.geom_id < nMeshes ? base[i.geom_id] : 0u;
    if (inst.empty())
        inst.resize(1);
    return dev_upload(c, c->dShadeInst, inst.data(), sizeof(skh_instance) * inst.size());
}

// the box around the baked light proxies' group from its bounds {lo xyz, hi xyz} (a radiance ray that misses it skips the proxies' tree)
static void set_light_box(skh_context* c, bool any, const float* b6)
{
    // What a visit of the group's root could find must lie inside: encode_node4's margin -- the LARGEST coordinate magnitude over all three axes
    // times 2^-20, the same on every axis (a flat light on the plane x = 0 still gets it: ADVICE r5) -- plus two quantisation cells of the
    // root node per axis (child planes are rounded outward to the node's power-of-two grid, cell < 2 extent / 255).  A larger box only sends
    // a few more rays to the proxies' tree.
    float mAll = 0.0f;
    for (int k = 0; k < 3; ++k)
        mAll = std::max(mAll, std::max(std::fabs(b6[k]), std::fabs(b6[3 + k])));
    for (int k = 0; k < 3; ++k)
    {
        const float ext = (b6[3 + k] + mAll * 0x1p-20f) - (b6[k] - mAll * 0x1p-20f);
        const float m = any ? mAll * 0x1p-20f + 1e-30f + 2.0f * (2.0f * ext / 255.0f) : 0.0f;
        c->lightBox.lo[k] = any ? b6[k] - m : -INFINITY;
        c->lightBox.hi[k] = any ? b6[3 + k] + m : INFINITY;
    }
}

skh_status skh_build_accel(skh_context* c, uint32_t flags)
{
    if (!c)
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    (void)hipSetDevice(c->device);
    {
        const skh_status ss = build_shading_tables(c);
        if (ss != SKH_OK)
            return ss;
    }
    for (uint32_t i = 0; i < c->nInstances; ++i)
    {
        // what the kernels index with an instance's ids: a mesh / curve set (traversal, shading) and, for light proxies, a light
        const skh_instance& in = c->instances[i];
        const char* what = nullptr;
        if (in.type > SKH_INSTANCE_CURVE)
            what = "type";
        else if (in.type == SKH_INSTANCE_CURVE ? in.geom_id >= c->curves.size() : in.geom_id >= c->meshes.size())
            what = "geom_id";
        else if (in.type == SKH_INSTANCE_LIGHT && in.light_id >= c->nLights)
            what = "light_id";
        if (what)
        {
            c->err = "skh_build_accel: instance " + std::to_string(i) + " has an out-of-range " + what;
            return SKH_INVALID_ARGUMENT;
        }
    }
    const bool usePloc = (flags & SKH_BUILD_SAH) != 0 || c->buildQuality != 0;
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t st = c->stream;
    skh_status s;
    const uint32_t B = 256;
    const uint32_t nMeshes = (uint32_t)c->meshes.size();
    const uint32_t nInst = c->nInstances;
    // ---- instance validity (finite inverse) and the baked set ----
    std::vector<float> w2o(12 * (size_t)std::max(1u, nInst));
    std::vector<uint8_t> valid(std::max(1u, nInst));
    for (uint32_t i = 0; i < nInst; ++i)
        valid[i] = invert_affine(c->instances[i].transform, &w2o[12 * (size_t)i]) ? 1 : 0;
    std::vector<uint32_t> meshUsers(nMeshes, 0u), meshUsersLeft(nMeshes, 0u); // users: mesh + light instances; left: those that keep their TLAS leaf
    c->baked.assign(std::max(1u, nInst), 0);
    // two world-space groups: [0] mesh instances (every ray), [1] light proxies (radiance rays only: shadow rays do not see lights,
    // RAY_MASK_SHADOW, closest_hit.cu:191) -- the ray mask stays a property of the group, not of the triangle
    std::vector<uint32_t> wInst, wFirst;
    bool worldCurves = false;
    std::vector<uint32_t> worldCurveInst; // curve instances with a table entry (and their set's tree) of their own
    std::vector<std::vector<uint32_t>> mergedGroups; // ... and the groups of equal transforms whose segments share one tree each
    uint32_t nBakedG[2] = { 0, 0 };
    {
        uint64_t uniqueTris = 0;
        for (const skh_mesh& me : c->meshes)
            uniqueTris += me.index_count / 3;
        for (uint32_t i = 0; i < nInst; ++i)
            if (c->instances[i].type != SKH_INSTANCE_CURVE && valid[i]) // (a singular instance is disabled: the reference's distant-light proxy is one, of mesh 0, scene.cpp:337-345)
                meshUsers[c->instances[i].geom_id]++;
        const uint64_t smallBudget = std::max<uint64_t>(uniqueTris, 1ull << 20);
        uint64_t total = 0;
        std::vector<uint8_t> pick(std::max(1u, nInst), 0);
        auto trisOf = [&](uint32_t i) { return c->meshes[c->instances[i].geom_id].index_count / 3; };
        auto eligible = [&](uint32_t i) { return c->bakeWorld >= 1 && c->instances[i].type != SKH_INSTANCE_CURVE && valid[i] && trisOf(i) > 0; };
        // mode 4 (default) = everything (3) while the instanced triangles stay within bake_budget_mtris million, else 2: a scene
        // whose instancing fits in memory many times over is fastest with no instancing at all (kitchen stand-in, 23 M instanced
        // triangles = 1.8 GB of leaf records + nodes of 288 GB: closest-hit 98.0 -> 91.0 ms), a forest of 10^9 instanced triangles is not
        uint32_t mode = c->bakeWorld;
        if (mode == 4)
        {
            uint64_t all = 0;
            for (uint32_t i = 0; i < nInst; ++i)
                all += eligible(i) ? trisOf(i) : 0u;
            mode = all <= (uint64_t)c->bakeBudgetMTris * 1000000ull ? 3u : 2u;
        }
        for (uint32_t i = 0; i < nInst; ++i)
            pick[i] = eligible(i) && (meshUsers[c->instances[i].geom_id] == 1 || mode >= 3) ? 1 : 0;
        if (mode == 2)
        {
            // Small SHARED meshes (boards, quads) follow only when that leaves no mesh instance behind: beside a populated top level
            // a partly baked scene measured no faster (kitchen stand-in, any-hit 41.1 -> 42.6 ms), an emptied one saves the level
            bool all = true;
            uint64_t sum = 0;
            for (uint32_t i = 0; i < nInst; ++i)
                if (c->instances[i].type == SKH_INSTANCE_MESH && eligible(i) && !pick[i])
                {
                    all = all && trisOf(i) <= c->bakeSmallTris;
                    sum += trisOf(i);
                }
            all = all && sum <= smallBudget;
            for (uint32_t i = 0; i < nInst; ++i)
                if (eligible(i) && !pick[i] && trisOf(i) <= c->bakeSmallTris && (all || c->instances[i].type == SKH_INSTANCE_LIGHT))
                    pick[i] = 1;
        }
        for (uint32_t i = 0; i < nInst; ++i)
            if (pick[i])
            {
                if (total + trisOf(i) >= (1ull << 27)) // (leaf references address 2^28 primitives)
                    pick[i] = 0;
                else
                    total += trisOf(i);
            }
        // Light proxies follow the meshes only when that EMPTIES the top level (then no ray ever leaves world space): beside a
        // populated TLAS their own group costs every radiance ray a root visit that the TLAS's distance-ordered culling mostly
        // avoided (kitchen stand-in, closest-hit 99.4 -> 102.4 ms), without a TLAS it saves the whole level (unshared variant: 89.7 -> 81.2 ms)
        // "World curves" (round 5): when a scene holds at most SKH_WORLD_CURVES curve instances with segments (and nothing else needs a top
        // level), their trees are walked straight from the world-only kernel -- no TLAS, no instance-entry pass; the instance's transform is
        // still applied to the ray, by the same operations --, so they do not keep the top level alive either.  An integer rule the CPU
        // checker evaluates too (it decides which light proxies are baked).
        bool meshStays = false;
        std::vector<uint32_t> curveInst;
        for (uint32_t i = 0; i < nInst; ++i)
        {
            const skh_instance& in = c->instances[i];
            if (in.type == SKH_INSTANCE_CURVE)
            {
                uint32_t segs = 0; // (a curve set without a segment has no BLAS and no leaf)
                if (in.geom_id < c->curves.size())
                    for (uint32_t k = 0; k < c->curves[in.geom_id].vertex_counts_count; ++k)
                        segs += std::max(3u, c->curveVertexCounts[c->curves[in.geom_id].vertex_counts_start + k]) - 3u;
                if (valid[i] && segs > 0)
                    curveInst.push_back(i);
            }
            else if (in.type == SKH_INSTANCE_MESH)
                meshStays = meshStays || (valid[i] && !pick[i] && c->meshes[in.geom_id].index_count >= 3);
        }
        bool lightsAllPicked = true; // (a light proxy the bake mode leaves behind keeps the top level alive as well)
        for (uint32_t i = 0; i < nInst; ++i)
            if (c->instances[i].type == SKH_INSTANCE_LIGHT && eligible(i) && !pick[i])
                lightsAllPicked = false;
        // (round 6) Curve instances under the SAME transform, bit for bit -- a groom handed over as several HdBasisCurves rprims, under the identity or
        // under the one Xform the whole character sits under -- do not take a table entry each: the segments of such a group are MERGED into one
        // curve tree in the group's object space (the first groups of the curve build; the segment records name their instance), which takes ONE
        // entry and is entered through the group's one transform.  Walking 8 / 16 per-prim trees one after the other cost the hair stand-in 33 % /
        // 50 % of its rate (each ray visits every tree's root and whatever overlaps: 40 / 53 instead of 28 nodes per radiance ray), docs/LOG.md.
        std::vector<std::vector<uint32_t>> xformGroups; // curve instances by transform, in order of first appearance
        for (uint32_t i : curveInst)
        {
            size_t g = 0;
            for (; g < xformGroups.size(); ++g)
                if (memcmp(c->instances[xformGroups[g][0]].transform, c->instances[i].transform, sizeof(float) * 12) == 0)
                    break;
            if (g == xformGroups.size())
                xformGroups.emplace_back();
            xformGroups[g].push_back(i);
        }
        worldCurves = !meshStays && !curveInst.empty() && xformGroups.size() <= SKH_WORLD_CURVES && lightsAllPicked;
        if (worldCurves)
        {
            // option curve_merge = 0 (A/B, tests): every instance keeps a tree and a table entry of its own while the table can hold them all
            const bool merge = c->curveMerge || curveInst.size() > SKH_WORLD_CURVES;
            if (merge)
                mergedGroups = xformGroups;
            else
                worldCurveInst = curveInst;
        }
        const bool tlasStays = meshStays || (!curveInst.empty() && !worldCurves);
        for (uint32_t i = 0; i < nInst; ++i)
        {
            const skh_instance& in = c->instances[i];
            if (in.type == SKH_INSTANCE_CURVE)
                continue;
            if (in.type == SKH_INSTANCE_LIGHT && tlasStays)
                pick[i] = 0;
            if (!pick[i])
                meshUsersLeft[in.geom_id] += valid[i] ? 1u : 0u;
        }
        total = 0;
        for (uint32_t g = 0; g < 2; ++g)
            for (uint32_t i = 0; i < nInst; ++i)
            {
                const skh_instance& in = c->instances[i];
                if (!pick[i] || (in.type == SKH_INSTANCE_LIGHT ? 1u : 0u) != g)
                    continue;
                c->baked[i] = 1;
                valid[i] = 0; // no TLAS leaf
                wInst.push_back(i);
                wFirst.push_back((uint32_t)total);
                const uint32_t nt = c->meshes[in.geom_id].index_count / 3;
                total += nt;
                nBakedG[g] += nt;
            }
        wFirst.push_back((uint32_t)total);
    }
    const uint32_t nBaked = nBakedG[0] + nBakedG[1];
    c->nBakedTris = nBaked;
    c->nBakedInst = (uint32_t)wInst.size();
    // ---- triangles of all meshes (a mesh whose users were all baked needs no object-space tree) + the baked group ----
    std::vector<uint32_t> triMesh, triLocal, meshTriCount(nMeshes + 2u);
    for (uint32_t m = 0; m < nMeshes; ++m)
    {
        const uint32_t nt = (meshUsers[m] > 0 && meshUsersLeft[m] == 0) ? 0u : c->meshes[m].index_count / 3;
        meshTriCount[m] = nt;
        for (uint32_t t = 0; t < nt; ++t)
        {
            triMesh.push_back(m);
            triLocal.push_back(t);
        }
    }
    // (merge_light_proxies: one world-space group for both kinds -- a light proxy is told from a mesh triangle by its primitive word, which
    // lacks SKH_PRIM_DIRECT (k_gather_tris); any-hit queries, which must not see lights (RAY_MASK_SHADOW, closest_hit.cu:191), skip those triangles)
    const bool mergeLights = c->mergeLightProxies != 0u && nBakedG[0] != 0u && nBakedG[1] != 0u;
    const uint32_t nGroup0 = mergeLights ? nBakedG[0] + nBakedG[1] : nBakedG[0], nGroup1 = mergeLights ? 0u : nBakedG[1];
    meshTriCount[nMeshes] = nGroup0;
    meshTriCount[nMeshes + 1u] = nGroup1;
    const uint32_t nMeshTris = (uint32_t)triMesh.size();
    const uint32_t nTris = nMeshTris + nBaked;
    c->nTris = nTris;
    LbvhOut triOut, segOut;
    DevBuf dTriMesh, dTriLocal, dBoxLo, dBoxHi, dGrp, dSegStart, dSegCurve, dSegLocal, dSegBuildStart, dSegInstOf, dW2o, dValid, dWInst, dWFirst;
    auto cleanup = [&]() {
        for (DevBuf* b : { &dTriMesh, &dTriLocal, &dBoxLo, &dBoxHi, &dGrp, &dSegStart, &dSegCurve, &dSegLocal, &dSegBuildStart, &dSegInstOf, &dW2o, &dValid, &dWInst, &dWFirst,
                           &triOut.sortedVals, &segOut.sortedVals, &triOut.groupRoot, &segOut.groupRoot, &triOut.groupBounds,
                           &segOut.groupBounds })
            dev_free(*b);
    };
#define BA(expr)                    \
    if ((s = (expr)) != SKH_OK)     \
    {                               \
        cleanup();                  \
        return s;                   \
    }
    BA(dev_upload(c, dTriMesh, triMesh.data(), sizeof(uint32_t) * (size_t)nMeshTris));
    BA(dev_upload(c, dTriLocal, triLocal.data(), sizeof(uint32_t) * (size_t)nMeshTris));
    BA(dev_upload(c, dWInst, wInst.data(), sizeof(uint32_t) * wInst.size()));
    BA(dev_upload(c, dWFirst, wFirst.data(), sizeof(uint32_t) * wFirst.size()));
    BA(dev_alloc(c, dBoxLo, sizeof(float4) * (size_t)std::max(1u, nTris)));
    BA(dev_alloc(c, dBoxHi, sizeof(float4) * (size_t)std::max(1u, nTris)));
    BA(dev_alloc(c, dGrp, sizeof(uint32_t) * (size_t)std::max(1u, nTris)));
    if (nMeshTris)
        k_tri_boxes<<<(nMeshTris + B - 1) / B, B, 0, st>>>(c->dVerts.as<uint8_t>(), c->dIndices.as<uint32_t>(), c->dMeshes.as<uint4>(),
                                                          dTriMesh.as<uint32_t>(), dTriLocal.as<uint32_t>(), nMeshTris, dBoxLo.as<float4>(),
                                                          dBoxHi.as<float4>(), dGrp.as<uint32_t>());
    if (nBaked)
        k_baked_tri_boxes<<<(nBaked + B - 1) / B, B, 0, st>>>(c->dInstances.as<uint8_t>(), dWInst.as<uint32_t>(), dWFirst.as<uint32_t>(),
                                                             (uint32_t)wInst.size(), c->dVerts.as<uint8_t>(), c->dIndices.as<uint32_t>(),
                                                             c->dMeshes.as<uint4>(), nBaked, nMeshTris, nMeshes, nGroup0, dBoxLo.as<float4>(),
                                                             dBoxHi.as<float4>(), dGrp.as<uint32_t>());
    const uint32_t riMin = c->reinsertMinSize ? c->reinsertMinSize : (nTris <= (4u << 20) ? 1u : 32u);
    BA(lbvh_build(c, nTris, nMeshes + 2u, meshTriCount, dBoxLo.as<float4>(), dBoxHi.as<float4>(), dGrp.as<uint32_t>(), (int)c->leafMaxTris, usePloc, triOut,
                  0, c->leafLines ? 48u : 0u, usePloc ? c->reinsertRounds : 0u, riMin));
    memset(&c->buildInfo, 0, sizeof(c->buildInfo));
    c->buildInfo.triangles = nTris;
    c->buildInfo.nodes = triOut.numNodes;
    c->hierNodes = triOut.numNodes;
    c->buildInfo.reinsert_rounds = triOut.riRounds;
    c->buildInfo.reinsert_moves = triOut.riMoves;
    c->buildInfo.reinsert_min_size = triOut.riMinSize;
    c->buildInfo.cost_before = triOut.riCostBefore;
    c->buildInfo.cost_after = triOut.riRounds ? triOut.riCostAfter : triOut.riCostBefore;
    c->buildInfo.ms_reinsert = triOut.riMs;
    const uint32_t nTriSlots = triOut.numSlots; // >= nTris: the line layout pads in front of leaves that would straddle a 128-byte line
    {
        // Which word a baked mesh triangle's hit carries.  The index of its shading record (k_shade fetches the record beside the instance's: one short
        // step less) -- unless a 16-byte hit record {t, u, v, instance << B | primitive} (compact_hits) has room for the mesh-local index only:
        // a closest-hit lane's second scattered store costs more than that step (kitchen closest-hit -4 %).
        uint32_t maxMeshTris = 1;
        for (uint32_t m = 0; m < nMeshes; ++m)
            maxMeshTris = std::max(maxMeshTris, c->meshes[m].index_count / 3u);
        auto fits = [&](uint32_t range) {
            uint32_t B = 1;
            while (B < 31u && (1u << B) < range)
                ++B;
            return (1ull << B) >= range && (uint64_t)nInst <= (1ull << (32u - B)) - 1ull;
        };
        const uint32_t nSegAll = (uint32_t)std::min<uint64_t>(0x7fffffffull, [&]() { uint64_t n = 0; for (const skh_curve& cu : c->curves) for (uint32_t k = 0; k < cu.vertex_counts_count; ++k) { const uint32_t ncp = c->curveVertexCounts[cu.vertex_counts_start + k]; n += ncp > 3u ? ncp - 3u : 0u; } return n; }());
        const uint32_t rangeDirect = std::max(c->nShadeRecords, nSegAll), rangeLocal = std::max(maxMeshTris, nSegAll);
        c->directRecords = c->directRecordsOpt >= 0 ? (uint32_t)c->directRecordsOpt : ((c->compactHits && !fits(rangeDirect) && fits(rangeLocal)) ? 0u : 1u);
        c->hitPrimRange = c->directRecords ? rangeDirect : rangeLocal;
    }
    c->nTriSlots = nTriSlots;
    BA(dev_alloc(c, c->dTris, sizeof(float4) * 3 * (size_t)std::max(1u, nTriSlots)));
    if (nTriSlots)
        k_gather_tris<<<(nTriSlots + B - 1) / B, B, 0, st>>>(c->dVerts.as<uint8_t>(), c->dIndices.as<uint32_t>(), c->dMeshes.as<uint4>(),
                                                              dTriMesh.as<uint32_t>(), dTriLocal.as<uint32_t>(),
                                                              triOut.sortedVals.as<uint32_t>(), nTriSlots, nMeshTris, c->dInstances.as<uint8_t>(), c->dShadeInst.as<uint8_t>(),
                                                              dWInst.as<uint32_t>(), dWFirst.as<uint32_t>(), (uint32_t)wInst.size(), c->directRecords,
                                                              c->dTris.as<float4>());
    dev_free(c->dTriNodes);
    c->dTriNodes = triOut.nodes;
    // what skh_refit_accel needs of this build: k_gather_tris' tables and the tree's levels (the old ones go out with this call's temporaries)
    std::swap(c->dTriOrder, triOut.sortedVals);
    std::swap(c->dTriMeshK, dTriMesh);
    std::swap(c->dTriLocalK, dTriLocal);
    std::swap(c->dWInstK, dWInst);
    std::swap(c->dWFirstK, dWFirst);
    c->triLevelStart = triOut.levelStart;
    c->nMeshTrisBuilt = nMeshTris, c->nWInstBuilt = (uint32_t)wInst.size(), c->triNumNodes = triOut.numNodes, c->lastBuildFlags = flags, c->nGroup1Built = nGroup1;
    c->builtGeomSig = c->geomSig, c->builtNVerts = c->nVerts;
    c->refitReady = true; // (either builder ends in the same collapse)
    c->worldRoot = nGroup0 ? triOut.hostGroupRoot[nMeshes] : SKH_REF_INVALID;
    c->lightRoot = nGroup1 ? triOut.hostGroupRoot[nMeshes + 1u] : SKH_REF_INVALID;
    float worldBounds[6] = { INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY };
    for (int k = 0; k < 3; ++k)
        c->lightBox.lo[k] = -INFINITY, c->lightBox.hi[k] = INFINITY;
    if (nBaked)
    {
        float gb[12];
        if (hipMemcpyAsync(gb, triOut.groupBounds.as<float>() + 6 * (size_t)nMeshes, sizeof(gb), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
        {
            c->err = "skh_build_accel: bounds of the baked groups: read-back failed";
            cleanup();
            return SKH_FAIL;
        }
        set_light_box(c, nGroup1 != 0u, gb + 6);
        for (uint32_t g = 0; g < 2; ++g)
            if (g == 0 ? nGroup0 : nGroup1)
                for (int k = 0; k < 3; ++k)
                {
                    worldBounds[k] = std::min(worldBounds[k], gb[6 * g + k]);
                    worldBounds[3 + k] = std::max(worldBounds[3 + k], gb[6 * g + 3 + k]);
                }
    }
    // ---- curve segments of all curve sets (segment enumeration: OptixRender.cpp:226-245) ----
    const uint32_t nCurves = (uint32_t)c->curves.size();
    std::vector<uint32_t> setSegStart, curveSegCount(nCurves), curveSegBase(nCurves); // per set, set after set: what k_shade looks a hit's control points up in
    for (uint32_t ci = 0; ci < nCurves; ++ci)
    {
        const skh_curve& cu = c->curves[ci];
        curveSegBase[ci] = (uint32_t)setSegStart.size();
        uint32_t off = 0, local = 0;
        for (uint32_t k = 0; k < cu.vertex_counts_count; ++k)
        {
            const uint32_t ncp = c->curveVertexCounts[cu.vertex_counts_start + k];
            for (int i = 0; i < (int)ncp - 3; ++i, ++local)
                setSegStart.push_back(cu.points_start + off + (uint32_t)i);
            off += ncp;
        }
        curveSegCount[ci] = local;
    }
    for (uint32_t k = 0; k < (uint32_t)setSegStart.size(); ++k)
        if ((uint64_t)setSegStart[k] + 4 > c->nPoints)
        {
            c->err = "skh_build_accel: curve segment reads past the control-point buffer";
            cleanup();
            return SKH_INVALID_ARGUMENT;
        }
    // Is anything else going to keep a top-level leaf?  (By the rule above nothing does when worldCurves holds; checked, not assumed: a curve instance
    // without a leaf AND without the world-only kernel would vanish.)  Decided HERE, before the curve build and before the validity flags go to the device
    // (the host-sweep TLAS path picks its leaves by the device's flags: ADVICE r5).
    bool othersInTlas = false;
    for (uint32_t i = 0; i < nInst; ++i)
        if (valid[i] && c->instances[i].type != SKH_INSTANCE_CURVE)
        {
            const std::vector<int>& roots = triOut.hostGroupRoot;
            othersInTlas = othersInTlas || (c->instances[i].geom_id < roots.size() && roots[c->instances[i].geom_id] != SKH_REF_INVALID);
        }
    const bool worldCurveKernel = worldCurves && c->worldKernel && !othersInTlas;
    if (!worldCurveKernel)
        mergedGroups.clear(); // (every curve instance keeps its TLAS leaf and its set's own tree)
    // The curve build's primitives, group after group: groups [0, SKH_WORLD_CURVES) = the segments of the MERGED transform groups (object space of the
    // group's transform; the record names the instance), group SKH_WORLD_CURVES + s = curve set s (skipped when every instance of the set was merged).
    std::vector<uint32_t> segStart, segCurve, segLocal, segInstOf;
    std::vector<uint32_t> curveGroupCount(nCurves + SKH_WORLD_CURVES, 0u);
    {
        std::vector<uint8_t> setNeeded(nCurves, mergedGroups.empty() ? 1 : 0);
        if (!mergedGroups.empty())
        {
            std::vector<uint8_t> isMerged(std::max(1u, nInst), 0);
            for (const auto& grp : mergedGroups)
                for (uint32_t i : grp)
                    isMerged[i] = 1;
            for (uint32_t i = 0; i < nInst; ++i)
                if (c->instances[i].type == SKH_INSTANCE_CURVE && !isMerged[i] && c->instances[i].geom_id < nCurves)
                    setNeeded[c->instances[i].geom_id] = 1;
        }
        for (size_t g = 0; g < mergedGroups.size(); ++g)
            for (uint32_t i : mergedGroups[g])
            {
                const uint32_t ci = c->instances[i].geom_id;
                for (uint32_t l = 0; l < curveSegCount[ci]; ++l)
                {
                    segStart.push_back(setSegStart[curveSegBase[ci] + l]);
                    segCurve.push_back((uint32_t)g);
                    segLocal.push_back(l);
                    segInstOf.push_back(i);
                }
                curveGroupCount[g] += curveSegCount[ci];
            }
        for (uint32_t ci = 0; ci < nCurves; ++ci)
            if (setNeeded[ci])
            {
                for (uint32_t l = 0; l < curveSegCount[ci]; ++l)
                {
                    segStart.push_back(setSegStart[curveSegBase[ci] + l]);
                    segCurve.push_back(SKH_WORLD_CURVES + ci);
                    segLocal.push_back(l);
                    segInstOf.push_back(0xffffffffu);
                }
                curveGroupCount[SKH_WORLD_CURVES + ci] = curveSegCount[ci];
            }
    }
    const uint32_t nSegs = (uint32_t)segStart.size();
    c->nSegs = nSegs;
    BA(dev_upload(c, c->dSegStartAll, setSegStart.data(), sizeof(uint32_t) * setSegStart.size()));
    BA(dev_upload(c, c->dCurveSegBase, curveSegBase.data(), sizeof(uint32_t) * (size_t)nCurves));
    BA(dev_upload(c, dSegBuildStart, segStart.data(), sizeof(uint32_t) * (size_t)nSegs));
    BA(dev_upload(c, dSegInstOf, segInstOf.data(), sizeof(uint32_t) * (size_t)nSegs));
    BA(dev_upload(c, dSegCurve, segCurve.data(), sizeof(uint32_t) * (size_t)nSegs));
    BA(dev_upload(c, dSegLocal, segLocal.data(), sizeof(uint32_t) * (size_t)nSegs));
    const bool segNode = c->curveSegNode != 0 && nSegs > 0;
    const uint32_t K = segNode ? 1u : std::max(1u, std::min(c->curveSplit, 8u)); // (segment nodes: the tree's primitives are whole segments)
    if ((uint64_t)nSegs * K >= (1ull << 28) || (segNode && (uint64_t)nSegs * 2 >= (1ull << 28)))
    {
        c->err = "skh_build_accel: more than 2^28 curve sub-segments (lower the curve_split option)";
        cleanup();
        return SKH_INVALID_ARGUMENT;
    }
    const uint32_t nSub = nSegs * K;
    const uint32_t nCurveGroups = nCurves + SKH_WORLD_CURVES; // [0, SKH_WORLD_CURVES) the merged transform groups, [SKH_WORLD_CURVES + s] curve set s
    std::vector<uint32_t> curveSubCount(nCurveGroups);
    for (uint32_t g = 0; g < nCurveGroups; ++g)
        curveSubCount[g] = curveGroupCount[g] * K;
    BA(dev_alloc(c, dBoxLo, sizeof(float4) * (size_t)std::max(nSub, std::max(1u, c->nInstances))));
    BA(dev_alloc(c, dBoxHi, sizeof(float4) * (size_t)std::max(nSub, std::max(1u, c->nInstances))));
    BA(dev_alloc(c, dGrp, sizeof(uint32_t) * (size_t)std::max(nSub, std::max(1u, c->nInstances))));
    if (nSub && segNode)
        k_seg_union_boxes<<<(nSegs + B - 1) / B, B, 0, st>>>(c->dPoints.as<float>(), c->dRadii.as<float>(), dSegBuildStart.as<uint32_t>(),
                                                            dSegCurve.as<uint32_t>(), nSegs, dBoxLo.as<float4>(), dBoxHi.as<float4>(), dGrp.as<uint32_t>());
    else if (nSub)
        k_seg_boxes<<<(nSub + B - 1) / B, B, 0, st>>>(c->dPoints.as<float>(), c->dRadii.as<float>(), dSegBuildStart.as<uint32_t>(),
                                                      dSegCurve.as<uint32_t>(), nSub, K, dBoxLo.as<float4>(), dBoxHi.as<float4>(),
                                                      dGrp.as<uint32_t>());
    BA(lbvh_build(c, nSub, nCurveGroups, curveSubCount, dBoxLo.as<float4>(), dBoxHi.as<float4>(), dGrp.as<uint32_t>(), segNode ? 1 : (int)c->curveLeaf, usePloc, segOut, 2, 0u,
                  usePloc ? c->reinsertCurveRounds : 0u, c->reinsertMinSize ? c->reinsertMinSize : (nSub <= (8u << 20) ? 1u : 32u)));
    const uint32_t strandMajor = segNode && c->curveStrandMajor ? 1u : 0u;
    if (segNode)
    {
        // the tree's nodes, then one segment node per segment; every one-segment leaf reference (internal nodes' child words, the per-set roots) is
        // redirected to the segment node in front of the leaf
        DevBuf grown;
        BA(dev_alloc(c, grown, sizeof(Node4) * ((size_t)segOut.numNodes + nSegs + 1)));
        if (segOut.numNodes)
            SKH_TRY(c, hipMemcpyAsync(grown.p, segOut.nodes.p, sizeof(Node4) * (size_t)segOut.numNodes, hipMemcpyDeviceToDevice, st));
        k_segnode_emit<<<(nSegs + B - 1) / B, B, 0, st>>>(c->dPoints.as<float>(), c->dRadii.as<float>(), dSegBuildStart.as<uint32_t>(), segOut.sortedVals.as<uint32_t>(), nSegs,
                                                         strandMajor, grown.as<Node4>() + segOut.numNodes);
        if (segOut.numNodes)
            k_segnode_patch<<<(segOut.numNodes * 4u + B - 1) / B, B, 0, st>>>(reinterpret_cast<int*>(grown.p) + 12, segOut.numNodes * 4u, 16u, 4u, segOut.sortedVals.as<uint32_t>(),
                                                                             strandMajor, segOut.numNodes);
        k_segnode_patch<<<(nCurveGroups + B - 1) / B, B, 0, st>>>(segOut.groupRoot.as<int>(), nCurveGroups, 1u, 1u, segOut.sortedVals.as<uint32_t>(), strandMajor, segOut.numNodes);
        SKH_TRY(c, hipMemcpyAsync(segOut.hostGroupRoot.data(), segOut.groupRoot.p, sizeof(int) * (size_t)nCurveGroups, hipMemcpyDeviceToHost, st));
        SKH_TRY(c, hipStreamSynchronize(st));
        dev_free(segOut.nodes);
        segOut.nodes = grown;
        segOut.numNodes += nSegs;
    }
    BA(dev_alloc(c, c->dSegs, sizeof(float4) * SKH_SEG_STRIDE * (size_t)std::max(1u, nSub)));
    if (nSub)
        k_gather_segs<<<(nSub + B - 1) / B, B, 0, st>>>(c->dPoints.as<float>(), c->dRadii.as<float>(), dSegBuildStart.as<uint32_t>(),
                                                        dSegLocal.as<uint32_t>(), dSegInstOf.as<uint32_t>(), segOut.sortedVals.as<uint32_t>(), nSub, K,
                                                        c->dSegs.as<float4>(), strandMajor);
    c->curveSplitBuilt = K;
    dev_free(c->dSegNodes);
    c->dSegNodes = segOut.nodes;
    c->hierNodes += segOut.numNodes;
    // what skh_refit_accel needs of the curve build (segment nodes carry boxes of their own kind: no refit for that experimental build)
    std::swap(c->dSegOrder, segOut.sortedVals);
    std::swap(c->dSegBuildStartK, dSegBuildStart);
    std::swap(c->dSegLocalK, dSegLocal);
    std::swap(c->dSegInstOfK, dSegInstOf);
    c->segLevelStart = segOut.levelStart;
    c->nSubBuilt = nSub, c->segNumNodes = segOut.numNodes;
    c->builtCurveSig = c->curveSig, c->curvePointsEdited = false;
    c->curveRefitReady = !segNode;
    // world curves: walked from the world-only kernel (option world_kernel; without it they keep their TLAS leaves: same hit records).  Decided BEFORE the
    // validity flags go to the device: both TLAS builders then agree on which instances have a leaf.
    std::vector<int> curveSetRoot(nCurves, SKH_REF_INVALID); // root of curve set s's own tree (group SKH_WORLD_CURVES + s)
    for (uint32_t ci = 0; ci < nCurves; ++ci)
        curveSetRoot[ci] = segOut.hostGroupRoot[SKH_WORLD_CURVES + ci];
    c->numWorldCurves = 0;
    c->worldCurveIdentLast = 0;
    c->worldCurveMerged = 0;
    c->numMergedCurveInst = 0;
    if (worldCurveKernel)
    {
        // (an entry under a bit-exact identity transform goes LAST in the table = first off every ray's stack: the kernel skips its matrix fetch)
        static const float kIdentity[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
        for (size_t k = 0; k + 1 < worldCurveInst.size(); ++k)
            if (memcmp(c->instances[worldCurveInst[k]].transform, kIdentity, sizeof(kIdentity)) == 0)
            {
                std::swap(worldCurveInst[k], worldCurveInst.back());
                break;
            }
        for (uint32_t i : worldCurveInst)
        {
            const int root = c->instances[i].geom_id < nCurves ? curveSetRoot[c->instances[i].geom_id] : SKH_REF_INVALID;
            if (root == SKH_REF_INVALID)
                continue;
            c->worldCurveRoot[c->numWorldCurves] = root;
            c->worldCurveInst[c->numWorldCurves++] = i;
            c->worldCurveIdentLast = memcmp(c->instances[i].transform, kIdentity, sizeof(kIdentity)) == 0 ? 1u : 0u; // (of the last one entered)
            valid[i] = 0; // no TLAS leaf
        }
        std::vector<size_t> order; // merged groups: the identity group last
        for (size_t g = 0; g < mergedGroups.size(); ++g)
            if (memcmp(c->instances[mergedGroups[g][0]].transform, kIdentity, sizeof(kIdentity)) != 0)
                order.push_back(g);
        for (size_t g = 0; g < mergedGroups.size(); ++g)
            if (memcmp(c->instances[mergedGroups[g][0]].transform, kIdentity, sizeof(kIdentity)) == 0)
                order.push_back(g);
        for (size_t g : order)
        {
            for (uint32_t i : mergedGroups[g])
                valid[i] = 0;
            if (segOut.hostGroupRoot[g] == SKH_REF_INVALID)
                continue;
            c->worldCurveMerged |= 1u << c->numWorldCurves; // = "the segment record names the instance" (DevScene::segInst); the entry's instance lends its transform
            c->worldCurveRoot[c->numWorldCurves] = segOut.hostGroupRoot[g];
            c->worldCurveInst[c->numWorldCurves++] = mergedGroups[g][0];
            c->worldCurveIdentLast = memcmp(c->instances[mergedGroups[g][0]].transform, kIdentity, sizeof(kIdentity)) == 0 ? 1u : 0u;
            c->numMergedCurveInst += (uint32_t)mergedGroups[g].size();
        }
    }
    // ---- instances -> TLAS (baked instances were marked invalid above: they get a record for shading, no leaf) ----
    BA(dev_upload(c, dW2o, w2o.data(), sizeof(float) * w2o.size()));
    BA(dev_upload(c, dValid, valid.data(), valid.size()));
    BA(dev_alloc(c, c->dDevInst, sizeof(DevInstance) * (size_t)std::max(1u, nInst)));
    if (nInst)
        k_instance_boxes<<<(nInst + B - 1) / B, B, 0, st>>>(c->dInstances.as<HostInstance>(), dW2o.as<float>(), dValid.as<uint8_t>(),
                                                           triOut.groupBounds.as<float>(), triOut.groupRoot.as<int>(),
                                                           segOut.groupBounds.as<float>() + 6 * SKH_WORLD_CURVES, segOut.groupRoot.as<int>() + SKH_WORLD_CURVES /* curve set s = group SKH_WORLD_CURVES + s */, nMeshes, nCurves,
                                                           nInst, c->dDevInst.as<DevInstance>(), dBoxLo.as<float4>(),
                                                           dBoxHi.as<float4>(), dGrp.as<uint32_t>());
    if (nInst > 0 && c->tightInstanceBoxes)
        k_instance_tight_boxes<<<nInst, 256, 0, st>>>(c->dInstances.as<HostInstance>(), c->dDevInst.as<DevInstance>(), c->dMeshes.as<uint4>(),
                                                      c->dVerts.as<uint8_t>(), nMeshes, 1u << 22, dBoxLo.as<float4>(), dBoxHi.as<float4>());
    uint32_t nValidHost = 0;
    for (uint32_t i = 0; i < nInst; ++i)
        nValidHost += valid[i] ? 1u : 0u;
    const bool tlasOnGpu = c->tlasBuild == 1 || (c->tlasBuild == 2 && nValidHost > 8192u);
    if (nInst > 0 && tlasOnGpu && c->tlasOpen <= 1)
    {
        // ---- TLAS on the GPU: the same PLOC + 4-wide collapse that builds the BLASes, over the instance boxes.  Which instances
        //      take part is known on the host without reading anything back: a finite inverse and a non-empty BLAS. ----
        std::vector<uint32_t> leafInst;
        for (uint32_t i = 0; i < nInst; ++i)
        {
            const skh_instance& in = c->instances[i];
            const std::vector<int>& roots = in.type == SKH_INSTANCE_CURVE ? curveSetRoot : triOut.hostGroupRoot;
            if (valid[i] && in.geom_id < roots.size() && roots[in.geom_id] != SKH_REF_INVALID)
                leafInst.push_back(i);
        }
        const uint32_t nLeaves = (uint32_t)leafInst.size();
        c->numTlasLeaves = nLeaves;
        dev_free(c->dTlasNodes);
        dev_free(c->dTlasInst);
        if (nLeaves == 0)
        {
            dev_free(c->dTravInst);
            c->tlasRoot = SKH_REF_INVALID;
        }
        else
        {
            DevBuf dLeafInst, dLeafLo, dLeafHi, dLeafGrp, dTinstTmp;
            LbvhOut tlasOut;
            auto cleanupT = [&]() {
                for (DevBuf* b : { &dLeafInst, &dLeafLo, &dLeafHi, &dLeafGrp, &dTinstTmp, &tlasOut.sortedVals, &tlasOut.groupRoot, &tlasOut.groupBounds })
                    dev_free(*b);
            };
#define BT(expr)                \
    if ((s = (expr)) != SKH_OK) \
    {                           \
        cleanupT();             \
        dev_free(tlasOut.nodes); \
        cleanup();              \
        return s;               \
    }
            BT(dev_upload(c, dLeafInst, leafInst.data(), sizeof(uint32_t) * (size_t)nLeaves));
            BT(dev_alloc(c, dLeafLo, sizeof(float4) * (size_t)nLeaves));
            BT(dev_alloc(c, dLeafHi, sizeof(float4) * (size_t)nLeaves));
            BT(dev_alloc(c, dLeafGrp, sizeof(uint32_t) * (size_t)nLeaves));
            BT(dev_alloc(c, dTinstTmp, sizeof(DevInstance) * (size_t)nLeaves));
            BT(dev_alloc(c, c->dTravInst, sizeof(DevInstance) * (size_t)nLeaves));
            k_tlas_leaves<<<(nLeaves + B - 1) / B, B, 0, st>>>(dLeafInst.as<uint32_t>(), nLeaves, c->dDevInst.as<DevInstance>(), dBoxLo.as<float4>(),
                                                               dBoxHi.as<float4>(), dLeafLo.as<float4>(), dLeafHi.as<float4>(), dLeafGrp.as<uint32_t>(),
                                                               dTinstTmp.as<DevInstance>());
            BT(lbvh_build(c, nLeaves, 1, std::vector<uint32_t>{ nLeaves }, dLeafLo.as<float4>(), dLeafHi.as<float4>(), dLeafGrp.as<uint32_t>(), 1, true, tlasOut, 1));
            // traversal records into leaf order (a TLAS leaf ref carries its position: sc.tinst + first)
            k_permute_instances<<<(nLeaves + B - 1) / B, B, 0, st>>>(dTinstTmp.as<DevInstance>(), tlasOut.sortedVals.as<uint32_t>(), nLeaves,
                                                                     c->dTravInst.as<DevInstance>());
            float gb[6];
            if (hipMemcpyAsync(gb, tlasOut.groupBounds.p, sizeof(gb), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            {
                cleanupT();
                dev_free(tlasOut.nodes);
                cleanup();
                c->err = "skh_build_accel: TLAS bounds read-back failed";
                return SKH_FAIL;
            }
            for (int k = 0; k < 3; ++k)
            {
                c->sceneLo[k] = gb[k];
                c->sceneHi[k] = gb[3 + k];
            }
            c->tlasRoot = tlasOut.hostGroupRoot[0];
            c->dTlasNodes = tlasOut.nodes;
            if (getenv("SKH_DEBUG"))
                fprintf(stderr, "[skh] TLAS (GPU PLOC): %u instances, %u leaves, %u nodes, root %d\n", nInst, nLeaves, tlasOut.numNodes, c->tlasRoot);
            cleanupT();
#undef BT
        }
    }
    else if (nInst > 0)
    {
        // instance boxes were produced on the device (k_instance_boxes); the sweep runs on the host
        std::vector<float4> hlo(nInst), hhi(nInst);
        std::vector<DevInstance> hinst(nInst);
        if (hipStreamSynchronize(st) != hipSuccess ||
            hipMemcpy(hlo.data(), dBoxLo.p, sizeof(float4) * nInst, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hhi.data(), dBoxHi.p, sizeof(float4) * nInst, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hinst.data(), c->dDevInst.p, sizeof(DevInstance) * nInst, hipMemcpyDeviceToHost) != hipSuccess)
        {
            c->err = "skh_build_accel: instance box read-back failed";
            cleanup();
            return SKH_FAIL;
        }
        std::vector<HostBox> hb;
        std::vector<uint32_t> ids;
        std::vector<DevInstance> tinst; // traversal records, one per TLAS leaf; pad = id of the instance it belongs to
        float slo[3] = { INFINITY, INFINITY, INFINITY }, shi[3] = { -INFINITY, -INFINITY, -INFINITY };
        {
            // ---- TLAS opening (partial re-braiding) ----
            // A TLAS leaf is (instance, BLAS subtree).  Starting from one leaf per instance, the leaf with the largest world
            // box is replaced by the children of its subtree's root until the budget is used: big instances that contain or
            // overlap others (rooms, floors, rotated boxes) stop dragging every ray through their whole box.  Results do not
            // depend on it (closest hit = min t, ties by (instance, primitive)); only nodes / instance entries per ray do.
            std::vector<Node4> hTri, hSeg;
            const bool open = c->tlasOpen > 1;
            if (open)
            {
                hTri.resize(triOut.numNodes);
                hSeg.resize(segOut.numNodes);
                if ((triOut.numNodes && hipMemcpy(hTri.data(), c->dTriNodes.p, sizeof(Node4) * hTri.size(), hipMemcpyDeviceToHost) != hipSuccess) ||
                    (segOut.numNodes && hipMemcpy(hSeg.data(), c->dSegNodes.p, sizeof(Node4) * hSeg.size(), hipMemcpyDeviceToHost) != hipSuccess))
                {
                    c->err = "skh_build_accel: BLAS node read-back failed";
                    cleanup();
                    return SKH_FAIL;
                }
            }
            struct Ref
            {
                uint32_t inst;
                int node;
                HostBox box;
                float area;
            };
            auto areaOf = [](const HostBox& b) { return hb_half_area(b.lo, b.hi); };
            auto cmp = [](const Ref& a, const Ref& b) { return a.area < b.area || (a.area == b.area && a.inst > b.inst); };
            std::vector<Ref> heap, done;
            uint32_t nValid = 0;
            for (uint32_t i = 0; i < nInst; ++i)
                if (hinst[i].mask != 0)
                {
                    Ref r{ i, hinst[i].rootRef, HostBox{ { hlo[i].x, hlo[i].y, hlo[i].z }, { hhi[i].x, hhi[i].y, hhi[i].z } }, 0.0f };
                    r.area = areaOf(r.box);
                    heap.push_back(r);
                    ++nValid;
                }
            std::make_heap(heap.begin(), heap.end(), cmp);
            const size_t budget = open ? (size_t)nValid * c->tlasOpen : (size_t)nValid;
            while (!heap.empty())
            {
                std::pop_heap(heap.begin(), heap.end(), cmp);
                const Ref r = heap.back();
                heap.pop_back();
                const std::vector<Node4>& bn = hinst[r.inst].type == 2 ? hSeg : hTri;
                if (!open || r.node < 0 || r.node == SKH_REF_INVALID || (size_t)r.node >= bn.size() || heap.size() + done.size() + 4 > budget)
                {
                    done.push_back(r);
                    continue;
                }
                const Node4& nd = bn[(size_t)r.node];
                const float* M = c->instances[r.inst].transform; // object -> world, 3x4 row-major
                for (int ch = 0; ch < 4; ++ch)
                {
                    if (nd.child[ch] == SKH_REF_INVALID)
                        continue;
                    // child box in object space (decoded as the traversal kernel sees it), padded, then the box of its
                    // eight transformed corners, padded again and clipped to the parent leaf's box
                    double lo[3], hi[3];
                    for (int a = 0; a < 3; ++a)
                    {
                        const double cell = (double)(a == 0 ? nd.cellx : (a == 1 ? nd.celly : nd.cellz));
                        lo[a] = (double)nd.o[a] + cell * (double)((nd.qlo[a] >> (8 * ch)) & 0xffu);
                        hi[a] = (double)nd.o[a] + cell * (double)((nd.qhi[a] >> (8 * ch)) & 0xffu);
                        const double pad = (std::fabs(lo[a]) + std::fabs(hi[a])) * 0x1p-20 + 1e-30;
                        lo[a] -= pad;
                        hi[a] += pad;
                    }
                    Ref q{ r.inst, nd.child[ch], HostBox{ { INFINITY, INFINITY, INFINITY }, { -INFINITY, -INFINITY, -INFINITY } }, 0.0f };
                    for (int k = 0; k < 8; ++k)
                    {
                        const double x = (k & 1) ? hi[0] : lo[0], y = (k & 2) ? hi[1] : lo[1], z = (k & 4) ? hi[2] : lo[2];
                        for (int a = 0; a < 3; ++a)
                        {
                            const double w = (double)M[4 * a] * x + (double)M[4 * a + 1] * y + (double)M[4 * a + 2] * z + (double)M[4 * a + 3];
                            const double pad = std::fabs(w) * 0x1p-20 + 1e-30;
                            q.box.lo[a] = std::min(q.box.lo[a], (float)(w - pad));
                            q.box.hi[a] = std::max(q.box.hi[a], (float)(w + pad));
                        }
                    }
                    for (int a = 0; a < 3; ++a)
                    {
                        q.box.lo[a] = std::max(q.box.lo[a], r.box.lo[a]);
                        q.box.hi[a] = std::min(q.box.hi[a], r.box.hi[a]);
                        if (q.box.hi[a] < q.box.lo[a])
                            q.box.hi[a] = q.box.lo[a];
                    }
                    q.area = areaOf(q.box);
                    heap.push_back(q);
                    std::push_heap(heap.begin(), heap.end(), cmp);
                }
            }
            // deterministic leaf numbering: by (instance, node)
            std::sort(done.begin(), done.end(), [](const Ref& a, const Ref& b) { return a.inst < b.inst || (a.inst == b.inst && a.node < b.node); });
            hb.resize(done.size());
            tinst.resize(std::max<size_t>(1, done.size()));
            for (size_t r = 0; r < done.size(); ++r)
            {
                hb[r] = done[r].box;
                ids.push_back((uint32_t)r);
                tinst[r] = hinst[done[r].inst];
                tinst[r].rootRef = done[r].node;
                tinst[r].pad = done[r].inst;
                for (int k = 0; k < 3; ++k)
                {
                    slo[k] = std::min(slo[k], hb[r].lo[k]);
                    shi[k] = std::max(shi[k], hb[r].hi[k]);
                }
            }
            c->numTlasLeaves = (uint32_t)done.size();
        }
        BA(dev_upload(c, c->dTravInst, tinst.data(), sizeof(DevInstance) * tinst.size()));
        std::vector<Node4> hnodes;
        std::vector<uint32_t> horder;
        c->tlasRoot = tlas_sah_build(hb, ids, hnodes, horder);
        if (getenv("SKH_DEBUG"))
        {
            fprintf(stderr, "[skh] TLAS: %u instances, %zu leaves, %zu nodes, root %d\n", nInst, ids.size(), hnodes.size(), c->tlasRoot);
            for (uint32_t i = 0; i < std::min((uint32_t)hb.size(), 8u); ++i)
                fprintf(stderr, "[skh]  leaf %u inst %u root %d box %g %g %g .. %g %g %g\n", i, tinst[i].pad, tinst[i].rootRef, hb[i].lo[0], hb[i].lo[1], hb[i].lo[2], hb[i].hi[0], hb[i].hi[1], hb[i].hi[2]);
        }
        dev_free(c->dTlasNodes);
        dev_free(c->dTlasInst);
        BA(dev_upload(c, c->dTlasNodes, hnodes.data(), sizeof(Node4) * hnodes.size()));
        BA(dev_upload(c, c->dTlasInst, horder.data(), sizeof(uint32_t) * horder.size()));
        for (int k = 0; k < 3; ++k)
        {
            c->sceneLo[k] = ids.empty() ? 0.0f : slo[k];
            c->sceneHi[k] = ids.empty() ? 1.0f : shi[k];
        }
    }
    else
    {
        dev_free(c->dTlasNodes);
        dev_free(c->dTlasInst);
        dev_free(c->dTravInst);
        c->tlasRoot = SKH_REF_INVALID;
    }
    if (nBaked)
        for (int k = 0; k < 3; ++k)
        {
            const bool none = c->tlasRoot == SKH_REF_INVALID;
            c->sceneLo[k] = none ? worldBounds[k] : std::min(c->sceneLo[k], worldBounds[k]);
            c->sceneHi[k] = none ? worldBounds[3 + k] : std::max(c->sceneHi[k], worldBounds[3 + k]);
        }
    if (getenv("SKH_DEBUG"))
        fprintf(stderr, "[skh] bake_world %u: %u of %u instances baked, %u + %u world-space triangles, roots %d %d; %u object-space triangles; TLAS root %d\n", c->bakeWorld,
                c->nBakedInst, nInst, nBakedG[0], nBakedG[1], c->worldRoot, c->lightRoot, nMeshTris, c->tlasRoot);
    hipError_t e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess)
    {
        c->err = std::string("skh_build_accel: ") + hipGetErrorString(e);
        return SKH_FAIL;
    }
    c->accelBuilt = true;
    c->msBuild = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SKH_OK;
#undef BA
}

// ---------------------------------------------------------------------------------------------------------------
// skh_refit_accel: after a VERTEX edit (skh_set_geometry with the same mesh table and index buffer) -- the triangle hierarchy keeps its topology, its
// leaf records are gathered again from the new vertices and its boxes are recomputed bottom-up, one launch per level (skh_bvh.h k_node4_refit_level).
// north_star's "SAH refit".  Offered for what a real bake is: every mesh instance baked to world space (no top level: the world-only kernels); anything
// else -- a top level, edited instances / curves / options, another topology -- falls back to the full rebuild, and the build info says which happened.
// ---------------------------------------------------------------------------------------------------------------
skh_status skh_refit_accel(skh_context* c)
{
    if (!c)
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    (void)hipSetDevice(c->device);
    // (curves: control points and radii may have changed -- skh_set_curves with the built sets' table and vertex counts --; their tree is refitted the same way)
    const bool curvesOk = c->curves.empty() ? c->nSubBuilt == 0 : (c->curveRefitReady && c->curveSig == c->builtCurveSig);
    const bool can = c->refitReady && c->geomSig == c->builtGeomSig && c->nVerts == c->builtNVerts && c->tlasRoot == SKH_REF_INVALID && curvesOk &&
                     (c->triNumNodes > 0 || c->segNumNodes > 0);
    c->buildInfo.refit = 0;
    if (!can)
        return skh_build_accel(c, c->lastBuildFlags);
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t st = c->stream;
    const uint32_t B = 256;
    skh_status s = build_shading_tables(c); // (normals, tangents, uvs may have changed with the positions)
    if (s != SKH_OK)
        return s;
    if (c->nTriSlots)
        k_gather_tris<<<(c->nTriSlots + B - 1) / B, B, 0, st>>>(c->dVerts.as<uint8_t>(), c->dIndices.as<uint32_t>(), c->dMeshes.as<uint4>(), c->dTriMeshK.as<uint32_t>(),
                                                                 c->dTriLocalK.as<uint32_t>(), c->dTriOrder.as<uint32_t>(), c->nTriSlots, c->nMeshTrisBuilt, c->dInstances.as<uint8_t>(),
                                                                 c->dShadeInst.as<uint8_t>(), c->dWInstK.as<uint32_t>(), c->dWFirstK.as<uint32_t>(), c->nWInstBuilt, c->directRecords,
                                                                 c->dTris.as<float4>());
    if ((s = dev_alloc(c, c->dTriNodeBox, sizeof(float4) * 2 * (size_t)std::max(1u, c->triNumNodes))) != SKH_OK)
        return s;
    for (size_t L = c->triLevelStart.size() > 1 ? c->triLevelStart.size() - 1 : 0; L-- > 0;)
    {
        const uint32_t first = c->triLevelStart[L], count = c->triLevelStart[L + 1] - first;
        if (count)
            k_node4_refit_level<<<(count + B - 1) / B, B, 0, st>>>(c->dTriNodes.as<Node4>(), c->dTriNodeBox.as<float4>(), first, count, c->dTris.as<float4>());
    }
    if (c->curvePointsEdited && c->nSubBuilt)
    {
        // the curve tree: leaf records (control points, bounding cylinders) gathered again from the new points, boxes level by level
        k_gather_segs<<<(c->nSubBuilt + B - 1) / B, B, 0, st>>>(c->dPoints.as<float>(), c->dRadii.as<float>(), c->dSegBuildStartK.as<uint32_t>(), c->dSegLocalK.as<uint32_t>(),
                                                               c->dSegInstOfK.as<uint32_t>(), c->dSegOrder.as<uint32_t>(), c->nSubBuilt, c->curveSplitBuilt, c->dSegs.as<float4>(), 0u);
        if ((s = dev_alloc(c, c->dSegNodeBox, sizeof(float4) * 2 * (size_t)std::max(1u, c->segNumNodes))) != SKH_OK)
            return s;
        for (size_t L = c->segLevelStart.size() > 1 ? c->segLevelStart.size() - 1 : 0; L-- > 0;)
        {
            const uint32_t first = c->segLevelStart[L], count = c->segLevelStart[L + 1] - first;
            if (count)
                k_node4_refit_level_curves<<<(count + B - 1) / B, B, 0, st>>>(c->dSegNodes.as<Node4>(), c->dSegNodeBox.as<float4>(), first, count, c->dSegs.as<float4>(), c->curveSplitBuilt);
        }
        c->curvePointsEdited = false;
    }
    // the two baked groups' bounds: the box around the light proxies (radiance rays that miss it skip their tree), the scene box
    DevBuf dRefs, dOut;
    const int refs[2] = { c->worldRoot, c->lightRoot };
    float gb[12];
    if ((s = dev_upload(c, dRefs, refs, sizeof(refs))) != SKH_OK || (s = dev_alloc(c, dOut, sizeof(gb))) != SKH_OK)
    {
        dev_free(dRefs), dev_free(dOut);
        return s;
    }
    k_ref_boxes<<<1, 64, 0, st>>>(dRefs.as<int>(), 2u, c->dTris.as<float4>(), c->dTriNodeBox.as<float4>(), dOut.as<float>()); // (INVALID roots give empty boxes)
    hipError_t e = hipMemcpyAsync(gb, dOut.p, sizeof(gb), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    dev_free(dRefs), dev_free(dOut);
    if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess)
    {
        c->err = std::string("skh_refit_accel: ") + hipGetErrorString(e);
        c->accelBuilt = false, c->refitReady = false;
        return SKH_FAIL;
    }
    set_light_box(c, c->lightRoot != SKH_REF_INVALID && c->nGroup1Built != 0u, gb + 6);
    for (int k = 0; k < 3; ++k)
    {
        c->sceneLo[k] = std::min(c->worldRoot != SKH_REF_INVALID ? gb[k] : INFINITY, c->lightRoot != SKH_REF_INVALID ? gb[6 + k] : INFINITY);
        c->sceneHi[k] = std::max(c->worldRoot != SKH_REF_INVALID ? gb[3 + k] : -INFINITY, c->lightRoot != SKH_REF_INVALID ? gb[9 + k] : -INFINITY);
    }
    c->accelBuilt = true;
    c->buildInfo.refit = 1;
    c->msRefit = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->buildInfo.ms_refit = c->msRefit;
    c->refits++;
    return SKH_OK;
}

skh_status skh_get_baked(skh_context* c, uint8_t* flags, uint32_t n_instances, uint32_t* out_baked_instances, uint32_t* out_baked_triangles)
{
    if (!c)
        return SKH_INVALID_ARGUMENT;
    if (!c->accelBuilt)
    {
        const skh_status s = skh_build_accel(c, SKH_BUILD_LBVH);
        if (s != SKH_OK)
            return s;
    }
    const bool on = true;
    if (flags)
        for (uint32_t i = 0; i < n_instances; ++i)
            flags[i] = (on && i < c->nInstances && i < c->baked.size()) ? c->baked[i] : 0;
    if (out_baked_instances)
        *out_baked_instances = on ? c->nBakedInst : 0u;
    if (out_baked_triangles)
        *out_baked_triangles = on ? c->nBakedTris : 0u;
    return SKH_OK;
}

static skh_status alloc_frame(skh_context* c)
{
    skh_status s;
    const uint32_t T = c->tileSize;
    if (!c->customTiles)
    {
        c->tileXY.clear();
        for (uint32_t y = 0; y < c->height; y += T)
            for (uint32_t x = 0; x < c->width; x += T)
            {
                c->tileXY.push_back(x);
                c->tileXY.push_back(y);
            }
    }
    c->numTiles = (uint32_t)(c->tileXY.size() / 2);
    c->numSlots = c->numTiles * T * T;
    // sub-frame batching: aim at ~64 M paths per wavefront pass.  The persistent trace kernels have a long tail (a few
    // rays walk 10x the average number of nodes); at 2 M rays per launch the tail is half of the kernel time, at 64 M it
    // is amortised (measured at 1080p, Mray/s: 1 sub-frame per pass 1104, 16: 1942, 64: 2055 before the kernel work; with
    // the final kernels 8: 3455, 16: 3579, 32: 3678).  ~270 B per path: 18 GB of 288 GB at the default.
    c->batchCapacity = c->subframeBatch ? c->subframeBatch : std::min(64u, std::max(1u, (1u << 27) / std::max(1u, c->numSlots)));
    const size_t N1 = std::max(1u, c->numSlots);
    const size_t N = N1 * c->batchCapacity;
    // queues are sharded SKH_SHARDS ways: a shard's region holds an eighth of the paths (rounded up to whole waves); k_raygen deals the
    // first queue out in equal runs, and a k_shade workgroup emits into the shard it read from, so no shard outgrows its region
    c->queueRegion = (uint32_t)((((N + SKH_SHARDS - 1) / SKH_SHARDS) + 63) & ~(size_t)63);
    const size_t NQ = (size_t)SKH_SHARDS * c->queueRegion;
#define AF(expr)                \
    if ((s = (expr)) != SKH_OK) \
        return s;
    AF(dev_upload(c, c->dTileXY, c->tileXY.data(), sizeof(uint32_t) * c->tileXY.size()));
    {
        // k_raygen's tables: valid (inside the image) slots per 512-slot block, as an exclusive prefix sum
        auto compact = [](uint32_t x) {
            x &= 0x55555555u;
            x = (x ^ (x >> 1)) & 0x33333333u;
            x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
            x = (x ^ (x >> 4)) & 0x00ff00ffu;
            x = (x ^ (x >> 8)) & 0x0000ffffu;
            return x;
        };
        const uint32_t blocks = (c->numSlots + 511u) / 512u;
        std::vector<uint32_t> base(std::max(1u, blocks), 0u);
        const uint32_t shift2 = 2 * c->tileShift, mask = (1u << shift2) - 1u;
        uint32_t total = 0;
        for (uint32_t b = 0; b < blocks; ++b)
        {
            base[b] = total;
            const uint32_t end = std::min(c->numSlots, (b + 1) * 512u);
            for (uint32_t slot = b * 512u; slot < end; ++slot)
            {
                const uint32_t tile = slot >> shift2, m = slot & mask;
                const uint32_t px = c->tileXY[2 * tile] + compact(m), py = c->tileXY[2 * tile + 1] + compact(m >> 1);
                total += (px < c->width && py < c->height) ? 1u : 0u;
            }
        }
        c->raygenBlocksPerSub = blocks;
        c->raygenValidPerSub = total;
        AF(dev_upload(c, c->dRaygenBase, base.data(), sizeof(uint32_t) * base.size()));
    }
    AF(dev_alloc(c, c->dAccum, sizeof(float4) * N1));
    AF(dev_alloc(c, c->dDiffuse, sizeof(float4) * N1));
    AF(dev_alloc(c, c->dSpecular, sizeof(float4) * N1));
    AF(dev_alloc(c, c->dDiffCnt, sizeof(uint16_t) * N1));
    AF(dev_alloc(c, c->dSpecCnt, sizeof(uint16_t) * N1));
    AF(dev_alloc(c, c->dSums, sizeof(float) * 11 * N1));
    AF(dev_alloc(c, c->dPath, sizeof(float) * SKH_PATH_FLOATS * N));
    AF(dev_alloc(c, c->dRayQ[0], sizeof(float) * 9 * NQ));
    AF(dev_alloc(c, c->dRayQ[1], sizeof(float) * 9 * NQ));
    AF(dev_alloc(c, c->dHits, sizeof(float) * 8 * NQ));
    AF(dev_alloc(c, c->dShadowQ, sizeof(float) * 9 * NQ));
    AF(dev_alloc(c, c->dContrib, sizeof(float4) * NQ));
    c->queueConstFilled = false;
    AF(dev_alloc(c, c->dCounts, sizeof(uint32_t) * (SKH_COUNT_STRIDE * SKH_SHARDS * 2 * SKH_MAX_LAUNCH_ROUNDS + 16 * SKH_FETCH_STRIDE * SKH_MAX_LAUNCH_ROUNDS)));
    c->traceBlocks = (uint32_t)c->numCUs * c->wavesPerCU;
    AF(dev_alloc(c, c->dOvf, sizeof(int) * (size_t)(SKH_STACK_OVF + SKH_TAIL_EXTRA) * (uint32_t)c->numCUs * std::max(std::max(c->wavesPerCU, c->wavesPerCUWorld), std::max(c->wavesPerCUShadow, c->wavesPerCUShadowWorld)) * SKH_TRACE_BLOCK));
    AF(dev_alloc(c, c->dOvf2, sizeof(int) * (size_t)(SKH_STACK_OVF + SKH_TAIL_EXTRA) * (uint32_t)c->numCUs * std::max(std::max(c->wavesPerCU, c->wavesPerCUWorld), std::max(c->wavesPerCUShadow, c->wavesPerCUShadowWorld)) * SKH_TRACE_BLOCK));
#undef AF
    SKH_TRY(c, hipMemsetAsync(c->dAccum.p, 0, sizeof(float4) * N1, c->stream));
    SKH_TRY(c, hipMemsetAsync(c->dDiffuse.p, 0, sizeof(float4) * N1, c->stream));
    SKH_TRY(c, hipMemsetAsync(c->dSpecular.p, 0, sizeof(float4) * N1, c->stream));
    SKH_TRY(c, hipMemsetAsync(c->dDiffCnt.p, 0, sizeof(uint16_t) * N1, c->stream));
    SKH_TRY(c, hipMemsetAsync(c->dSpecCnt.p, 0, sizeof(uint16_t) * N1, c->stream));
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    return SKH_OK;
}

skh_status skh_resize(skh_context* c, uint32_t width, uint32_t height)
{
    if (!c || width == 0 || height == 0)
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    (void)hipSetDevice(c->device);
    c->width = width;
    c->height = height;
    return alloc_frame(c);
}

skh_status skh_set_tiles(skh_context* c, uint32_t tile_size, const uint32_t* tile_xy, uint32_t n_tiles)
{
    if (!c || tile_size < 8 || tile_size > 256 || (tile_size & (tile_size - 1)))
        return SKH_INVALID_ARGUMENT;
    spec_drop(c);
    (void)hipSetDevice(c->device);
    c->tileSize = tile_size;
    c->tileShift = 0;
    while ((1u << c->tileShift) < tile_size)
        ++c->tileShift;
    c->customTiles = tile_xy != nullptr;
    if (tile_xy)
        c->tileXY.assign(tile_xy, tile_xy + 2 * (size_t)n_tiles);
    if (c->width && c->height)
        return alloc_frame(c);
    return SKH_OK;
}

// ---- timing helpers ----
static hipEvent_t next_event(skh_context* c)
{
    if (c->eventsUsed == c->eventPool.size())
    {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        c->eventPool.push_back(e);
    }
    return c->eventPool[c->eventsUsed++];
}
struct SpanGuard
{
    skh_context* c;
    int cls;
    hipEvent_t a = nullptr;
    hipStream_t st;
    SpanGuard(skh_context* c_, int cls_, hipStream_t st_ = nullptr) : c(c_), cls(cls_), st(st_ ? st_ : c_->stream)
    {
        c->launches[cls]++;
        if (c->timing)
        {
            a = next_event(c);
            (void)hipEventRecord(a, st);
        }
    }
    ~SpanGuard()
    {
        if (c->timing)
        {
            hipEvent_t b = next_event(c);
            (void)hipEventRecord(b, st);
            c->spans.push_back(TimedSpan{ cls, a, b });
        }
    }
};
static void harvest_spans(skh_context* c)
{
    for (const TimedSpan& s : c->spans)
    {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess)
        {
            c->msClass[s.cls] += ms;
            if (getenv("SKH_DEBUG_SPANS"))
                fprintf(stderr, "[span] class %d %.3f ms\n", s.cls, ms);
        }
    }
    c->spans.clear();
    c->eventsUsed = 0;
}

static DevScene make_dev_scene(const skh_context* c)
{
    DevScene sc;
    sc.tlasNodes = c->dTlasNodes.as<Node4>();
    sc.tlasInst = c->dTlasInst.as<uint32_t>();
    sc.tlasRoot = c->tlasRoot;
    sc.numWorldCurves = c->numWorldCurves;
    sc.worldCurveIdentLast = c->worldCurveIdentLast;
    sc.worldCurveMerged = c->worldCurveMerged;
    for (uint32_t k = 0; k < SKH_WORLD_CURVES; ++k)
        sc.worldCurveRoot[k] = k < c->numWorldCurves ? c->worldCurveRoot[k] : SKH_REF_INVALID, sc.worldCurveInst[k] = k < c->numWorldCurves ? c->worldCurveInst[k] : 0u;
    sc.numInstances = c->nInstances;
    sc.inst = c->dDevInst.as<DevInstance>();
    sc.tinst = c->dTravInst.as<DevInstance>();
    sc.triNodes = c->dTriNodes.as<Node4>();
    sc.tris = c->dTris.as<float4>();
    sc.segNodes = c->dSegNodes.as<Node4>();
    sc.segs = c->dSegs.as<float4>();
    sc.curveSplit = c->curveSplitBuilt;
    sc.worldRoot = c->worldRoot;
    sc.lightRoot = c->lightRoot;
    sc.instances = c->dShadeInst.as<HostInstance>();
    sc.shadeTris = c->dShadeTris.as<float4>();
    sc.verts = c->dVerts.as<uint8_t>();
    sc.indices = c->dIndices.as<uint32_t>();
    sc.meshes = c->dMeshes.as<uint4>();
    sc.curveSegBase = c->dCurveSegBase.as<uint32_t>();
    sc.segStartAll = c->dSegStartAll.as<uint32_t>();
    sc.cpoints = c->dPoints.as<float>();
    sc.cradii = c->dRadii.as<float>();
    sc.lights = c->dLights.as<Light>();
    sc.numLights = c->nLights;
    sc.materials = c->dMaterials.as<Material>();
    sc.hairConst = c->dHairConst.as<HairConst>();
    sc.numMaterials = c->nMaterials;
    sc.texels = c->dTexels.as<uint32_t>();
    sc.texDesc = c->dTexDesc.as<uint4>();
    sc.numTextures = c->nTextures;
    sc.profile = c->dStats.as<StatsDev>();
    sc.overflowFlag = c->hOverflow; // (hipHostMallocMapped memory: the host pointer is valid on the device)
    return sc;
}

static skh_status ensure_ready(skh_context* c)
{
    if (!c->accelBuilt)
    {
        skh_status s = skh_build_accel(c, SKH_BUILD_LBVH);
        if (s != SKH_OK)
            return s;
    }
    if (c->nMaterials == 0)
    {
        // material 0 = default.mdl::default_material (OptixRender.cpp:1090-1097)
        skh_material m;
        memset(&m, 0, sizeof(m));
        m.type = SKH_MAT_DIFFUSE;
        m.base_color[0] = m.base_color[1] = m.base_color[2] = 0.8f;
        skh_status s = skh_set_materials(c, &m, 1);
        if (s != SKH_OK)
            return s;
    }
    return SKH_OK;
}

// one launch of the persistent trace kernel over a sharded queue: picks the build (world-only / two-level / two-level + curves) and the grid
template <bool ANY, bool COUNT>
static void launch_trace(skh_context* c, const DevScene& sc, RayQ rq, const uint32_t* countPtr, uint32_t* fetch,
                         HitQ hq, PathS ps, const float4* contrib, hipStream_t st = nullptr)
{
    // scenes without curve instances run the build of the kernel that has no curve intersector in it (fewer VGPRs)
    const bool curveBuild = c->nSegs != 0;
    // (small overlapped passes, triangle scenes: a closest-hit wave asks for rays when 48 instead of 32 of its lanes are idle -- 1-spp 1080p call 3.11 -> 3.01 ms; 40 / 56: 3.04 / 3.04; lower
    // thresholds lose: 24 / 16 / 8: 3.20 / 3.29 / 3.44; the any-hit threshold stays: 32 / 56 of 64: 3.16 / 3.12 against 3.11 at 48.  An explicit fetch_min_closest applies everywhere.)
    const uint32_t fetchMinClosest = (c->gridOverride && !curveBuild && !c->fetchMinClosestSet) ? c->fetchMinClosestSmall : (curveBuild ? c->curveFetchMinClosest : c->fetchMinClosest);
    const uint32_t fetchMin = ANY ? (curveBuild ? c->curveFetchMinShadow : c->fetchMinShadow) : fetchMinClosest;
    const uint32_t nodeBreak = ANY ? (curveBuild ? c->curveNodeBreakShadow : c->nodeBreakShadow) : (curveBuild ? c->curveNodeBreakClosest : c->nodeBreakClosest);
    // (the world-only kernel with the curve block -- curve instances under identity transforms -- wants the triangle kernels' node-loop exits and
    // an earlier curve block: hair stand-in 1 862 -> 1 937 Mray/s with 32 / 28 / 32 against the two-level curve build's 20 / 20 / 48, gpurun_out/r5t)
    const bool wcv = c->nSegs && c->tlasRoot == SKH_REF_INVALID && c->numWorldCurves > 0 && c->worldKernel;
    const uint32_t nodeBreakW = wcv ? (ANY ? c->nodeBreakShadow : c->nodeBreakClosest) : nodeBreak;
    const uint32_t fm = fetchMin | ((wcv ? c->worldCurveMin : c->curveMin) << 8) | (nodeBreakW << 16) | (c->leafMin << 24);
    if (!st)
        st = c->stream;
    int* ovf = st == c->stream ? c->dOvf.as<int>() : c->dOvf2.as<int>(); // (two trace kernels may be in flight)
    StatsDev* sd = c->dStats.as<StatsDev>();
    const bool worldOnly = !c->nSegs && c->tlasRoot == SKH_REF_INVALID && (c->worldRoot != SKH_REF_INVALID || c->lightRoot != SKH_REF_INVALID) && c->worldKernel;
    const bool worldCurves = c->nSegs && c->tlasRoot == SKH_REF_INVALID && c->numWorldCurves > 0 && c->worldKernel; // the world-only kernel with the curve block
    // (the world-only triangle builds run 8 waves per SIMD)
    const uint32_t fullGrid = (ANY && !c->nSegs) ? (uint32_t)c->numCUs * (worldOnly ? c->wavesPerCUShadowWorld : c->wavesPerCUShadow)
                                                 : (worldOnly ? (uint32_t)c->numCUs * c->wavesPerCUWorld : c->traceBlocks);
    // (overlapped small passes: small_waves_* are shares of a 32-wave CU; a build that fits fewer waves -- the curve builds: 24 -- takes the same share of what it fits,
    // so that the two concurrent launches still fit side by side: hair 1-spp calls 651 -> ~680 Mray/s against an unscaled 20 / 12)
    const uint32_t scaledOverride = (uint32_t)((uint64_t)c->gridOverride * fullGrid / (32u * (uint32_t)c->numCUs));
    const uint32_t blocks = c->gridOverride ? std::max((uint32_t)c->numCUs, std::min(scaledOverride, fullGrid)) : fullGrid;
    // A hierarchy that fits the L2 many times over (Cornell: 30 triangles) makes rays so cheap that the launch is bound by the eight queue cursors'
    // atomics (~88 M per second and address): such scenes reserve 128 positions per atomic (Cornell 13.5 -> 14.8 Gray/s).  Larger scenes lose by it
    // (neighbouring rays are then traced at different times by one wave instead of together by neighbouring waves: kitchen -1 %, hair -2 %;
    // docs/LOG.md, round 5); only the world-only triangle builds carry the code.
    const uint32_t chunk = c->fetchChunk >= 0 ? (uint32_t)c->fetchChunk : (c->hierNodes <= 16384u ? 128u : 0u);
#ifdef SKH_TAIL_PROFILE
    (void)hipMemsetAsync(&sd->launchT0[ANY ? 1 : 0], 0xff, sizeof(unsigned long long), st);
#endif
    if (worldOnly && (c->tailSplit == 2 || (c->tailSplit == 1 && c->hierNodes > 16384u) || c->splitNow))
        k_trace<ANY, COUNT, false, true, true><<<blocks, SKH_TRACE_BLOCK, 0, st>>>(sc, rq, countPtr, fetch, fm, hq, ps, contrib, ovf, sd, c->lightBox, chunk);
    else if (worldOnly)
        // every instance is baked: the world-only build of the kernel (no instance entry, no object-space copy of the ray)
        k_trace<ANY, COUNT, false, true><<<blocks, SKH_TRACE_BLOCK, 0, st>>>(sc, rq, countPtr, fetch, fm, hq, ps, contrib, ovf, sd, c->lightBox, chunk);
    else if (worldCurves)
        k_trace<ANY, COUNT, true, true><<<blocks, SKH_TRACE_BLOCK, 0, st>>>(sc, rq, countPtr, fetch, fm, hq, ps, contrib, ovf, sd, c->lightBox, chunk);
    else if (c->nSegs)
        k_trace<ANY, COUNT, true><<<blocks, SKH_TRACE_BLOCK, 0, st>>>(sc, rq, countPtr, fetch, fm, hq, ps, contrib, ovf, sd, c->lightBox, chunk);
    else
        k_trace<ANY, COUNT, false><<<blocks, SKH_TRACE_BLOCK, 0, st>>>(sc, rq, countPtr, fetch, fm, hq, ps, contrib, ovf, sd, c->lightBox, chunk);
}

// After a synchronisation: did any traversal of the calls since the last check drop a stack entry (its result may miss hits)?
static skh_status check_stack_overflow(skh_context* c, const char* where)
{
    if (!*c->hOverflow)
        return SKH_OK;
    *c->hOverflow = 0u;
    c->stackOverflows++;
    c->err = std::string(where) + ": a traversal stack overflowed its " + std::to_string(SKH_STACK_LDS) + " LDS + " + std::to_string(SKH_STACK_OVF) +
             " global entries and dropped a subtree: hits may be missing (degenerate hierarchy?)";
    return SKH_FAIL;
}

// One wavefront pass: either one launch of p->samples_this_launch samples (batch = 1), or `batch` consecutive sub-frames
// of one sample each traced together (more rays per launch; results identical, see k_finalize_batch).
static skh_status render_one(skh_context* c, const skh_frame_params* p, uint32_t batch, void* d_image, bool trace = true, uint32_t finalFirst = 0,
                             uint32_t finalCount = 0xffffffffu, uint32_t pathSel = 0 /* 1: the second path-state buffer */, bool finalize = true,
                             hipStream_t finStream = nullptr /* where the accumulation step runs (default: the render stream) */)
{
    if (p->max_depth > 128 || p->samples_this_launch == 0)
    {
        c->err = "skh_render_subframe: max_depth must be <= 128 (MAX_BOUNCES, RandomSampler.h:35) and samples_this_launch >= 1";
        return SKH_INVALID_ARGUMENT;
    }
    hipStream_t st = c->stream;
    FrameP fp;
    memcpy(fp.viewToWorld, p->view_to_world, sizeof(fp.viewToWorld));
    memcpy(fp.clipToView, p->clip_to_view, sizeof(fp.clipToView));
    fp.subframeIndex = p->subframe_index;
    fp.samplesThisLaunch = p->samples_this_launch;
    fp.sppTotal = p->spp_total;
    fp.maxDepth = p->max_depth;
    fp.rectMethod = p->rect_light_sampling_method;
    memcpy(fp.exposure, p->exposure, sizeof(fp.exposure));
    fp.enableAccumulation = p->enable_accumulation;
    fp.debug = p->debug;
    fp.shadowTmin = p->shadow_ray_tmin;
    fp.materialTmin = p->material_ray_tmin;
    fp.width = c->width;
    fp.height = c->height;
    fp.tileSize = c->tileSize;
    fp.tileShift = c->tileShift;
    fp.numTiles = c->numTiles;
    fp.numSlots = c->numSlots;
    fp.batch = batch;
    fp.finalFirst = std::min(finalFirst, batch);
    fp.finalCount = std::min(finalCount, batch - fp.finalFirst);
    const DevScene sc = make_dev_scene(c);
    const uint32_t N = c->numSlots * c->batchCapacity; // plane stride of the path-state buffer
    const uint32_t NQ = SKH_SHARDS * c->queueRegion; // plane stride of every queue (rays, hits, shadow contributions)
    const uint32_t NP = c->numSlots * batch; // paths in this pass
    if (NP == 0)
        return SKH_OK;
    const uint32_t gridSlots = (c->numSlots + 255) / 256;
    const uint32_t* tiles = c->dTileXY.as<uint32_t>();
    PathS ps{ pathSel ? c->dPathB.as<float>() : c->dPath.as<float>(), pathSel ? c->pathBStride : N };
    RayQ rq[2] = { RayQ{ c->dRayQ[0].as<float>(), NQ, c->queueRegion }, RayQ{ c->dRayQ[1].as<float>(), NQ, c->queueRegion } };
    RayQ shq{ c->dShadowQ.as<float>(), NQ, c->queueRegion };
    HitQ hq{ c->dHits.as<float>(), NQ };
    {
        // 16-byte hit records where they are possible: the world-only triangle kernels (every mesh hit names its shading record) and an instance
        // count and a record count that share 32 bits with the all-ones word left for a miss
        const bool worldKernels = c->tlasRoot == SKH_REF_INVALID && c->worldKernel &&
                                  (c->nSegs ? c->numWorldCurves != 0u : (c->worldRoot != SKH_REF_INVALID || c->lightRoot != SKH_REF_INVALID));
        const uint32_t range = std::max(1u, c->hitPrimRange); // shading records (or mesh-local indices), light proxies' triangles, curve segments: skh_build_accel
        uint32_t B = 1;
        while (B < 31u && (1u << B) < range)
            ++B;
        hq.primBits = (c->compactHits && worldKernels && c->nShadeRecords != 0u && (uint64_t)c->nInstances <= (1ull << (32u - B)) - 1ull && (1ull << B) >= range) ? B : 0u;
        hq.direct = c->directRecords;
        hq.recClamp = c->nShadeRecords ? c->nShadeRecords - 1u : 0u;
    }
    HitQ nohq{ nullptr, 0 };
    {
        // tmin / tmax of every radiance ray and tmin of every shadow ray are constants of the frame parameters (OptixRender.cu:121-122, closest_hit.cu):
        // their queue planes are filled here, when the queues are new or a value has changed, and never written per ray (8 + 8 + 4 B per path and bounce)
        uint32_t bits[2];
        memcpy(&bits[0], &fp.materialTmin, 4);
        memcpy(&bits[1], &fp.shadowTmin, 4);
        if (!c->queueConstFilled || bits[0] != c->queueConstBits[0] || bits[1] != c->queueConstBits[1])
        {
            const uint32_t fb = (uint32_t)c->numCUs * 8u;
            for (int k = 0; k < 2; ++k)
            {
                k_fill_f32<<<fb, 256, 0, st>>>(rq[k].base + (size_t)6 * NQ, NQ, fp.materialTmin);
                k_fill_f32<<<fb, 256, 0, st>>>(rq[k].base + (size_t)7 * NQ, NQ, 1e16f);
            }
            k_fill_f32<<<fb, 256, 0, st>>>(shq.base + (size_t)6 * NQ, NQ, fp.shadowTmin);
            c->queueConstFilled = true;
            c->queueConstBits[0] = bits[0];
            c->queueConstBits[1] = bits[1];
        }
    }
    // dCounts: 260 queues (2 per bounce) x SKH_SHARDS queue-length words, then the ray-fetch cursors (8 per trace launch); every word
    // that is the target of atomics has a 128-byte line of its own (returning atomics on one line serialise at ~88 per microsecond)
    uint32_t* counts = c->dCounts.as<uint32_t>();
    const uint32_t QW = SKH_COUNT_STRIDE * SKH_SHARDS; // words per queue's lengths
    uint32_t* fetch = counts + QW * 2 * SKH_MAX_LAUNCH_ROUNDS;
    // shadow[b] on a second stream: it depends on shade[b] only, and so does closest[b+1]; each fills the other's tail.  Ray
    // sorting shares scratch buffers between the two and keeps everything on one stream.
    // overlap 1 (default): passes up to 32 M paths -- a rank's share of an N-GPU frame, the one-sub-frame-per-call pattern and its
    // speculative passes.  Up to 8 M paths both kernels run on reduced grids (16 + 16 waves per CU: +7 % at 2 M paths); above that on
    // full grids, where the second kernel's blocks simply take the slots the first one's finishing waves leave (1/8 share of a 1080p
    // frame: 24.26 -> 23.87 ms, 93.3 -> 94.9 % of 1/8 of the full frame; 1/4: 97.0 -> 97.8 %).  The full single-GPU frame (64 M paths
    // per pass) gains 0.5 % and stays on one stream so that its per-kernel hipEvent spans do not overlap.  overlap 2: always, reduced grids.
    const bool smallPass = NP <= (1u << 23) || c->overlap == 2;
    const bool useOverlap = (c->overlap == 2 || (c->overlap == 1 && NP <= (1u << 25))) && fp.debug != 1;
    const uint32_t rounds = fp.maxDepth;
    c->splitNow = c->tailSplit < 0 && smallPass && NP >= (1u << 17);
    // One sub-frame of one sample, accumulated in this call (the reference caller's pattern): its sums ARE the path's radiance and event word -- the batch kernel reads
    // them where they lie (0.0f + radiance, / 1.0f: the same operations), and the k_collect launch and its 11 planes of sums are not needed.
    const bool oneSampleDirect = batch == 1 && fp.samplesThisLaunch == 1 && finalize && fp.finalFirst == 0 && fp.finalCount == 1;
    for (uint32_t s = 0; trace && s < fp.samplesThisLaunch; ++s)
    {
        SKH_TRY(c, hipMemsetAsync(counts, 0, sizeof(uint32_t) * (QW * 2 * SKH_MAX_LAUNCH_ROUNDS + 16 * SKH_FETCH_STRIDE * (rounds + 1)), st));
        {
            SpanGuard g(c, KC_RAYGEN);
            k_raygen<<<c->raygenBlocksPerSub * fp.batch, 512, 0, st>>>(fp, tiles, s, rq[0], counts, ps, c->dRaygenBase.as<uint32_t>(),
                                                                       c->raygenBlocksPerSub, c->raygenValidPerSub);
        }
        for (uint32_t b = 0; b < rounds; ++b)
        {
            {
                SpanGuard g(c, KC_TRACE_CLOSEST);
                c->gridOverride = (useOverlap && smallPass) ? ((b == 0 && c->smallWavesFirst) ? c->smallWavesFirst : c->smallWavesClosest ? c->smallWavesClosest : (c->nSegs ? 16u : 20u)) * (uint32_t)c->numCUs : 0u;
                if (c->countTraversal)
                    launch_trace<false, true>(c, sc, rq[b & 1], counts + 2 * b * QW, fetch + 16 * b * SKH_FETCH_STRIDE, hq, ps, nullptr);
                else
                    launch_trace<false, false>(c, sc, rq[b & 1], counts + 2 * b * QW, fetch + 16 * b * SKH_FETCH_STRIDE, hq, ps, nullptr);
            }
            if (useOverlap && b > 0)
                (void)hipStreamWaitEvent(st, c->evShadow, 0); // shade[b] reads the radiance shadow[b-1] adds to and reuses its queue
            {
                SpanGuard g(c, KC_SHADE);
                // a shard holds at most an eighth of the pass's paths (rounded up to whole waves): SKH_SHARDS x that many workgroups,
                // workgroup b on shard b & 7; those past the end of their shard leave at once
                const uint32_t perShard = (((NP + SKH_SHARDS - 1u) / SKH_SHARDS) + 63u) & ~63u;
                const dim3 sg(SKH_SHARDS * ((perShard + SKH_SHADE_BLOCK - 1) / SKH_SHADE_BLOCK));
#define SKH_SHADE_LAUNCH(HAIRB)                                                                                                                      \
    k_shade<HAIRB><<<sg, SKH_SHADE_BLOCK, 0, st>>>(sc, fp, s, b, tiles, rq[b & 1], counts + 2 * b * QW, hq, ps, rq[(b + 1) & 1], counts + 2 * (b + 1) * QW, \
                                                   shq, c->dContrib.as<float4>(), counts + (2 * b + 1) * QW)
                if (c->hasHairMaterial)
                    SKH_SHADE_LAUNCH(true);
                else
                    SKH_SHADE_LAUNCH(false);
#undef SKH_SHADE_LAUNCH
            }
            {
                hipStream_t sst = useOverlap ? c->stream2 : st;
                if (useOverlap)
                {
                    (void)hipEventRecord(c->evShade, st);
                    (void)hipStreamWaitEvent(sst, c->evShade, 0);
                }
                {
                    SpanGuard g(c, KC_TRACE_SHADOW, sst);
                    c->gridOverride = (useOverlap && smallPass) ? ((b + 1 == rounds && (c->smallWavesLast || (!c->smallWavesShadow && !c->nSegs))) ? (c->smallWavesLast ? c->smallWavesLast : 20u) : c->smallWavesShadow ? c->smallWavesShadow : (c->nSegs ? 16u : 12u)) * (uint32_t)c->numCUs : 0u;
                    if (c->countTraversal)
                        launch_trace<true, true>(c, sc, shq, counts + (2 * b + 1) * QW, fetch + (16 * b + 8) * SKH_FETCH_STRIDE, nohq, ps, c->dContrib.as<float4>(), sst);
                    else
                        launch_trace<true, false>(c, sc, shq, counts + (2 * b + 1) * QW, fetch + (16 * b + 8) * SKH_FETCH_STRIDE, nohq, ps, c->dContrib.as<float4>(), sst);
                }
                c->gridOverride = 0;
                if (useOverlap)
                    (void)hipEventRecord(c->evShadow, sst);
            }
            if (fp.debug == 1)
                break;
        }
        if (useOverlap)
            (void)hipStreamWaitEvent(st, c->evShadow, 0);
        {
            SpanGuard g(c, KC_ACCUM);
            k_add_stats<<<1, 64, 0, st>>>(counts, rounds, c->dStats.as<StatsDev>());
            if (batch == 1 && !oneSampleDirect)
                k_collect<<<gridSlots, 256, 0, st>>>(fp, tiles, s, ps, c->dSums.as<float>());
        }
    }
    c->splitNow = false;
    hipStream_t fs = finStream ? finStream : st;
    if (finalize && (batch > 1 || oneSampleDirect))
    {
        SpanGuard g(c, KC_ACCUM, fs);
        k_finalize_batch<<<gridSlots, 256, 0, fs>>>(fp, tiles, ps, c->dAccum.as<float4>(), c->dDiffuse.as<float4>(), c->dSpecular.as<float4>(),
                                                    c->dDiffCnt.as<uint16_t>(), c->dSpecCnt.as<uint16_t>(), reinterpret_cast<float4*>(d_image));
    }
    else if (finalize)
    {
        SpanGuard g(c, KC_ACCUM, fs);
        k_finalize<<<gridSlots, 256, 0, fs>>>(fp, tiles, c->dSums.as<float>(), c->dAccum.as<float4>(), c->dDiffuse.as<float4>(),
                                              c->dSpecular.as<float4>(), c->dDiffCnt.as<uint16_t>(), c->dSpecCnt.as<uint16_t>(),
                                              reinterpret_cast<float4*>(d_image));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        c->err = std::string("render launch: ") + hipGetErrorString(e);
        return SKH_FAIL;
    }
    return SKH_OK;
}

skh_status skh_render_subframes(skh_context* c, const skh_frame_params* params, uint32_t n_subframes, void* d_image)
{
    if (!c || !params)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (c->width == 0)
    {
        c->err = "skh_render_subframe: call skh_resize first";
        return SKH_INVALID_ARGUMENT;
    }
    skh_status s = ensure_ready(c);
    if (s != SKH_OK)
        return s;
    if (n_subframes != 1)
        spec_drop(c); // (the path state is about to be reused)
    else
        spec_drop(c, true);
    skh_frame_params p = *params;
    // single-sample sub-frames are traced `batchCapacity` at a time when that gives the GPU more rays per launch
    const uint32_t cap = (p.samples_this_launch == 1 && p.debug != 1) ? c->batchCapacity : 1u;
    for (uint32_t k = 0; k < n_subframes;)
    {
        const uint32_t b = std::min(cap, n_subframes - k);
        if ((s = render_one(c, &p, b, d_image)) != SKH_OK)
            return s;
        p.subframe_index += p.samples_this_launch * b;
        k += b;
    }
    SKH_TRY(c, hipStreamSynchronize(c->stream)); // the reference's render() is synchronous (OptixRender.cpp:1012)
    if (c->timing)
        harvest_spans(c);
    return check_stack_overflow(c, "skh_render_subframes");
}

// true when b continues the frame a belongs to: everything equal except the sub-frame index, which advances by `step`
static bool same_frame(const skh_frame_params& a, const skh_frame_params& b, uint32_t step)
{
    skh_frame_params x = a, y = b;
    if (y.subframe_index != x.subframe_index + step)
        return false;
    x.subframe_index = y.subframe_index = 0;
    return memcmp(&x, &y, sizeof(x)) == 0;
}

// speculate_async: start tracing the pass that follows the one being consumed.  Nothing is in flight and c->stream is idle here.
static skh_status spec_launch_next(skh_context* c)
{
    skh_context::Speculation& sp = c->spec;
    if (!c->speculateAsync || !sp.valid || sp.nextInFlight || c->timing)
        return SKH_OK; // (per-kernel hipEvent spans are harvested at every call: timing runs keep the synchronous scheme)
    const uint32_t nextStart = sp.params.subframe_index + sp.count;
    if (nextStart >= sp.params.spp_total)
        return SKH_OK;
    const uint32_t cap = std::min(c->speculateMax, c->batchCapacity);
    const uint32_t n = std::min(std::min(std::max(2u, sp.count * c->speculateGrow), cap), sp.params.spp_total - nextStart);
    const uint32_t other = sp.buf ^ 1u;
    if (other == 1u)
    {
        const uint32_t stride = c->numSlots * cap;
        skh_status s = dev_alloc(c, c->dPathB, sizeof(float) * SKH_PATH_FLOATS * (size_t)stride);
        if (s != SKH_OK)
            return s;
        c->pathBStride = stride;
    }
    SKH_TRY(c, hipMemcpy(sp.statsMark, c->dStats.p, sizeof(sp.statsMark), hipMemcpyDeviceToHost));
    sp.nextParams = sp.params;
    sp.nextParams.subframe_index = nextStart;
    sp.nextCount = n;
    skh_status s = render_one(c, &sp.nextParams, n, nullptr, true, 0, 0, other, false);
    if (s != SKH_OK)
        return s;
    sp.nextInFlight = true;
    return SKH_OK;
}

skh_status skh_render_subframe(skh_context* c, const skh_frame_params* params, void* d_image)
{
    if (!c || !params)
        return SKH_INVALID_ARGUMENT;
    skh_context::Speculation& sp = c->spec;
    const bool eligible = c->speculateMax > 1 && params->samples_this_launch == 1 && params->debug == 0 && c->width != 0 && c->batchCapacity > 1;
    if (!eligible)
    {
        spec_drop(c);
        return skh_render_subframes(c, params, 1, d_image);
    }
    (void)hipSetDevice(c->device);
    skh_status s;
    if (sp.nextInFlight && sp.valid && sp.consumed >= sp.count && same_frame(sp.nextParams, *params, 0))
    {
        // the caller has collected the whole pass and continues into the one in flight: wait for it, make it the current pass
        unsigned long long after[2] = { 0, 0 };
        SKH_TRY(c, hipStreamSynchronize(c->stream));
        SKH_TRY(c, hipMemcpy(after, c->dStats.p, sizeof(after), hipMemcpyDeviceToHost));
        sp.nextInFlight = false;
        sp.passRadiance = after[0] - sp.statsMark[0];
        sp.passShadow = after[1] - sp.statsMark[1];
        sp.buf ^= 1u;
        sp.params = sp.nextParams;
        sp.count = sp.nextCount;
        sp.consumed = 0;
        sp.lastBatch = sp.nextCount;
        if ((s = check_stack_overflow(c, "skh_render_subframe")) != SKH_OK)
            return s;
    }
    if (sp.valid && sp.consumed < sp.count && same_frame(sp.params, *params, sp.consumed))
    {
        // this sub-frame was traced ahead: its radiances wait in the path state, only its accumulation step is left (beside a pass in
        // flight it runs on its own stream)
        hipStream_t fin = sp.nextInFlight ? c->stream3 : c->stream;
        if ((s = render_one(c, &sp.params, sp.count, d_image, false, sp.consumed, 1, sp.buf, true, fin)) != SKH_OK)
            return s;
        sp.consumed++;
        sp.last = *params;
        sp.haveLast = true;
        sp.streak++;
        SKH_TRY(c, hipStreamSynchronize(fin));
        if (c->timing && !sp.nextInFlight)
            harvest_spans(c);
        if (sp.consumed == 1)
            return spec_launch_next(c); // (first delivery of a pass that was itself traced in flight)
        return SKH_OK;
    }
    // a fresh pass.  How far ahead: nothing on the first call of a frame or after any change (interactive camera motion restarts at
    // sub-frame 0 every call and must not pay for samples it will throw away); doubling while the caller keeps continuing the frame
    const bool continues = sp.haveLast && same_frame(sp.last, *params, 1);
    sp.streak = continues ? sp.streak + 1 : 0;
    uint32_t ahead = 1;
    if (continues && params->spp_total > params->subframe_index)
        ahead = std::min(std::min(std::max(2u, sp.lastBatch * c->speculateGrow), std::min(c->speculateMax, c->batchCapacity)), params->spp_total - params->subframe_index);
    spec_drop(c, true);
    if ((s = ensure_ready(c)) != SKH_OK)
        return s;
    if (ahead <= 1)
    {
        sp.lastBatch = 1;
        sp.last = *params;
        sp.haveLast = true;
        return skh_render_subframes(c, params, 1, d_image);
    }
    unsigned long long before[2] = { 0, 0 }, after[2] = { 0, 0 }; // (StatsDev starts with raysRadiance, raysShadow; the stream is idle here)
    SKH_TRY(c, hipMemcpy(before, c->dStats.p, sizeof(before), hipMemcpyDeviceToHost));
    if ((s = render_one(c, params, ahead, d_image, true, 0, 1)) != SKH_OK)
        return s;
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    SKH_TRY(c, hipMemcpy(after, c->dStats.p, sizeof(after), hipMemcpyDeviceToHost));
    sp.passRadiance = after[0] - before[0];
    sp.passShadow = after[1] - before[1];
    sp.valid = true;
    sp.buf = 0;
    sp.params = *params;
    sp.count = ahead;
    sp.consumed = 1;
    sp.lastBatch = ahead;
    sp.last = *params;
    sp.haveLast = true;
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    if (c->timing)
        harvest_spans(c);
    if ((s = check_stack_overflow(c, "skh_render_subframe")) != SKH_OK)
        return s;
    return spec_launch_next(c);
}

skh_status skh_tonemap(skh_context* c, void* d_image, uint32_t width, uint32_t height, uint32_t type, const float exposure[3], float gamma)
{
    if (!c || !d_image || !exposure)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    const uint32_t n = width * height; // the reference's kernels test `> n` (Tonemappers.cu:20); this build uses >= n
    hipStream_t st = c->spec.nextInFlight ? c->stream3 : c->stream; // (beside a pass traced ahead: not behind it)
    k_tonemap<<<(n + 255) / 256, 256, 0, st>>>(reinterpret_cast<float4*>(d_image), n, type, exposure[0], exposure[1], exposure[2], gamma);
    SKH_TRY(c, hipStreamSynchronize(st));
    return SKH_OK;
}

skh_status skh_buffer_alloc(skh_context* c, size_t bytes, void** out)
{
    if (!c || !out)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    *out = nullptr;
    SKH_TRY(c, hipMalloc(out, bytes ? bytes : 16));
    SKH_TRY(c, hipMemset(*out, 0, bytes ? bytes : 16));
    return SKH_OK;
}
skh_status skh_buffer_free(skh_context* c, void* p)
{
    if (!c)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (p)
        SKH_TRY(c, hipFree(p));
    return SKH_OK;
}
skh_status skh_buffer_download(skh_context* c, const void* d, void* host, size_t bytes)
{
    if (!c || !d || !host)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (c->spec.nextInFlight)
    {
        // (the image was written by an accumulation step on stream3, which has completed; c->stream is busy with the pass traced
        // ahead and a null-stream copy would wait for it)
        SKH_TRY(c, hipMemcpyAsync(host, d, bytes, hipMemcpyDeviceToHost, c->stream3));
        SKH_TRY(c, hipStreamSynchronize(c->stream3));
        return SKH_OK;
    }
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    SKH_TRY(c, hipMemcpy(host, d, bytes, hipMemcpyDeviceToHost));
    return SKH_OK;
}

skh_status skh_host_register(skh_context* c, void* host, size_t bytes)
{
    if (!c || !host || !bytes)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    SKH_TRY(c, hipHostRegister(host, bytes, hipHostRegisterDefault));
    return SKH_OK;
}
skh_status skh_host_unregister(skh_context* c, void* host)
{
    if (!c || !host)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    SKH_TRY(c, hipHostUnregister(host));
    return SKH_OK;
}

static skh_status detile_to(skh_context* c, const DevBuf& src, void* d_dst)
{
    k_detile<<<(c->numSlots + 255) / 256, 256, 0, c->stream>>>(src.as<float4>(), c->dTileXY.as<uint32_t>(), c->numSlots, c->tileShift,
                                                              c->width, c->height, reinterpret_cast<float4*>(d_dst));
    return SKH_OK;
}

static skh_status read_slots(skh_context* c, const DevBuf& src, float* host)
{
    if (!c || !host || !c->width)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    const size_t bytes = sizeof(float4) * (size_t)c->width * c->height;
    skh_status s = dev_alloc(c, c->dScratchImage, bytes);
    if (s != SKH_OK)
        return s;
    SKH_TRY(c, hipMemsetAsync(c->dScratchImage.p, 0, bytes, c->stream));
    detile_to(c, src, c->dScratchImage.p);
    SKH_TRY(c, hipMemcpyAsync(host, c->dScratchImage.p, bytes, hipMemcpyDeviceToHost, c->stream));
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    return SKH_OK;
}
skh_status skh_read_accum(skh_context* c, float* host_rgba)
{
    return read_slots(c, c->dAccum, host_rgba);
}
skh_status skh_read_aov(skh_context* c, uint32_t which, float* host_rgba)
{
    if (!c)
        return SKH_INVALID_ARGUMENT;
    return read_slots(c, which == 0 ? c->dDiffuse : c->dSpecular, host_rgba);
}
skh_status skh_copy_accum(skh_context* c, void* d_dst)
{
    if (!c || !d_dst)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    detile_to(c, c->dAccum, d_dst);
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    return SKH_OK;
}
// the diffuse / specular AOV accumulators to a device image: what render() hands back once all samples are done and the debug view
// is 2 / 3 (OptixRender.cpp:1029-1042: cudaMemcpy(params.image, params.diffuse | params.specular))
skh_status skh_copy_aov(skh_context* c, uint32_t which, void* d_dst)
{
    if (!c || !d_dst || which > 1u)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (c->width == 0)
    {
        c->err = "skh_copy_aov: call skh_resize first";
        return SKH_INVALID_ARGUMENT;
    }
    detile_to(c, which == 0 ? c->dDiffuse : c->dSpecular, d_dst);
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    return SKH_OK;
}
skh_status skh_copy_accum_tiles(skh_context* c, void* d_dst)
{
    if (!c || !d_dst)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    SKH_TRY(c, hipMemcpyAsync(d_dst, c->dAccum.p, sizeof(float4) * (size_t)c->numSlots, hipMemcpyDeviceToDevice, c->stream));
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    return SKH_OK;
}
skh_status skh_scatter_tiles(skh_context* c, const void* d_src_tiles, const uint32_t* tile_xy, uint32_t n_tiles, uint32_t tile_size,
                             void* d_dst, uint32_t width, uint32_t height)
{
    if (!c || !d_src_tiles || !tile_xy || !d_dst || (tile_size & (tile_size - 1)))
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    // (tile list kept in a context buffer: no allocation per call; tiles whose origin lies outside the image -- the padding
    // of a gathered multi-rank tile set -- write nothing)
    skh_status s = dev_upload(c, c->dScatterXY, tile_xy, sizeof(uint32_t) * 2 * (size_t)n_tiles);
    if (s != SKH_OK)
        return s;
    uint32_t shift = 0;
    while ((1u << shift) < tile_size)
        ++shift;
    const uint32_t slots = n_tiles * tile_size * tile_size;
    k_detile<<<(slots + 255) / 256, 256, 0, c->stream>>>(reinterpret_cast<const float4*>(d_src_tiles), c->dScatterXY.as<uint32_t>(), slots, shift,
                                                        width, height, reinterpret_cast<float4*>(d_dst));
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
    {
        c->err = std::string("skh_scatter_tiles: ") + hipGetErrorString(e);
        return SKH_FAIL;
    }
    return SKH_OK;
}

// ---- multi-GPU tile gather (RCCL, loaded on first use) ----
namespace
{
struct RcclApi
{
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr; // optional
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr; // optional
    std::string why;
    bool overridden = false; // bound to SKH_RCCL_LIB instead of librccl (tests)
};
static bool rccl_load(RcclApi& api);
RcclApi* rccl()
{
    // (contexts on different GPUs may be driven from different threads: load once)
    static RcclApi api;
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] { ok = rccl_load(api); });
    return ok ? &api : nullptr;
}
static bool rccl_load(RcclApi& api)
{
    // SKH_RCCL_LIB (tests only): the library to bind instead -- tests/cpp/rccl_double.cpp, a stand-in for N ranks SHARING one GPU, so that the
    // N > 1 branch of skh_gather_tiles runs on a 1-GPU box.  RTLD_LOCAL: its nccl* symbols must not shadow the real ones PyTorch has loaded.
    if (const char* over = getenv("SKH_RCCL_LIB"))
    {
        if (!(api.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL)))
        {
            api.why = std::string("SKH_RCCL_LIB=") + over + ": " + dlerror();
            return false;
        }
        api.overridden = true;
    }
    // (a process that uses PyTorch has its librccl.so.1 loaded already: the same soname resolves to that copy)
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
        if (api.lib || (api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!api.lib)
    {
        api.why = "librccl.so.1 not found";
        return false;
    }
#define SKH_SYM(field, sym) (*(void**)(&api.field) = dlsym(api.lib, sym))
    SKH_SYM(GetUniqueId, "ncclGetUniqueId");
    SKH_SYM(CommInitRank, "ncclCommInitRank");
    SKH_SYM(CommDestroy, "ncclCommDestroy");
    SKH_SYM(Send, "ncclSend");
    SKH_SYM(Recv, "ncclRecv");
    SKH_SYM(GroupStart, "ncclGroupStart");
    SKH_SYM(GroupEnd, "ncclGroupEnd");
    SKH_SYM(GetErrorString, "ncclGetErrorString");
    SKH_SYM(CommCount, "ncclCommCount");
    SKH_SYM(CommUserRank, "ncclCommUserRank");
#undef SKH_SYM
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
    {
        api.why = "librccl.so.1 lacks the point-to-point API";
        api.lib = nullptr;
        return false;
    }
    return true;
}
} // namespace

skh_status skh_comm_unique_id(void* out_id)
{
    static_assert(sizeof(ncclUniqueId) == SKH_COMM_ID_BYTES, "id size");
    RcclApi* r = rccl();
    if (!out_id || !r)
        return out_id ? SKH_FAIL : SKH_INVALID_ARGUMENT;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess)
        return SKH_FAIL;
    memcpy(out_id, &id, sizeof(id));
    return SKH_OK;
}

skh_status skh_comm_init(skh_context* c, const void* id, int world_size, int rank)
{
    if (!c || !id || world_size < 1 || rank < 0 || rank >= world_size)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    RcclApi* r = rccl();
    if (!r)
    {
        c->err = "skh_comm_init: RCCL is not available (librccl.so.1)";
        return SKH_FAIL;
    }
    if (c->comm)
        (void)skh_comm_destroy(c);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    const ncclResult_t e = r->CommInitRank(&c->comm, world_size, uid, rank);
    if (e != ncclSuccess)
    {
        c->comm = nullptr;
        c->err = std::string("skh_comm_init: ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(e) : "error");
        return SKH_FAIL;
    }
    c->commWorld = world_size;
    c->commRank = rank;
    return SKH_OK;
}

skh_status skh_comm_info(skh_context* c, int* out_world, int* out_rank, int* out_rccl_ranks)
{
    if (!c)
        return SKH_INVALID_ARGUMENT;
    if (out_world)
        *out_world = c->commWorld;
    if (out_rank)
        *out_rank = c->commRank;
    if (out_rccl_ranks)
    {
        // what RCCL itself says about the communicator (ncclCommCount): 0 = no communicator (world size 1, or the caller gathers another way)
        int n = 0;
        RcclApi* r = c->comm ? rccl() : nullptr;
        if (r && r->CommCount && r->CommCount(c->comm, &n) != ncclSuccess)
            n = -1;
        *out_rccl_ranks = n;
    }
    return SKH_OK;
}

skh_status skh_comm_destroy(skh_context* c)
{
    if (!c)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (c->comm)
    {
        (void)hipStreamSynchronize(c->stream);
        RcclApi* r = rccl();
        if (r)
            (void)r->CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->commWorld = 1;
    c->commRank = 0;
    return SKH_OK;
}

skh_status skh_gather_tiles(skh_context* c, uint32_t max_tiles, void* d_recv, int root)
{
    if (!c || root < 0 || root >= c->commWorld || max_tiles < c->numTiles || (c->commRank == root && !d_recv))
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (c->commWorld > 1 && !c->comm)
    {
        c->err = "skh_gather_tiles: call skh_comm_init first";
        return SKH_INVALID_ARGUMENT;
    }
    const size_t tilePixels = (size_t)c->tileSize * c->tileSize;
    const size_t chunk = (size_t)max_tiles * tilePixels; // float4 per rank
    const size_t mine = (size_t)c->numTiles * tilePixels;
    const bool isRoot = c->commRank == root;
    // the root's own share goes straight into its slot of the receive buffer; a sender stages its tiles (zero-padded) once
    float4* dst = isRoot ? reinterpret_cast<float4*>(d_recv) + (size_t)root * chunk : nullptr;
    if (!isRoot)
    {
        skh_status s = dev_alloc(c, c->dTileSend, sizeof(float4) * chunk);
        if (s != SKH_OK)
            return s;
        dst = c->dTileSend.as<float4>();
    }
    SKH_TRY(c, hipMemcpyAsync(dst, c->dAccum.p, sizeof(float4) * mine, hipMemcpyDeviceToDevice, c->stream));
    if (chunk > mine)
        SKH_TRY(c, hipMemsetAsync(dst + mine, 0, sizeof(float4) * (chunk - mine), c->stream));
    if (c->commWorld > 1)
    {
        RcclApi* r = rccl();
        ncclResult_t e = r->GroupStart();
        if (isRoot)
        {
            for (int k = 0; k < c->commWorld && e == ncclSuccess; ++k)
                if (k != root)
                    e = r->Recv(reinterpret_cast<float4*>(d_recv) + (size_t)k * chunk, chunk * 4, ncclFloat, k, c->comm, c->stream);
        }
        else if (e == ncclSuccess)
            e = r->Send(dst, chunk * 4, ncclFloat, root, c->comm, c->stream);
        const ncclResult_t e2 = r->GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess)
        {
            c->err = std::string("skh_gather_tiles: ") + (r->GetErrorString ? r->GetErrorString(e != ncclSuccess ? e : e2) : "RCCL error");
            return SKH_FAIL;
        }
    }
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    return SKH_OK;
}

// ---- raw ray queries ----
// raw queries: ray k of the caller sits in shard k / per at offset k % per (per = rays per shard, a multiple of 64)
__global__ void k_rays_aos_to_soa(const skh_ray* __restrict__ rays, uint32_t n, uint32_t per, RayQ rq)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const skh_ray r = rays[k];
    const uint32_t g = k / per, i = g * rq.region + (k - g * per);
    rq.plane(0)[i] = r.origin[0];
    rq.plane(1)[i] = r.origin[1];
    rq.plane(2)[i] = r.origin[2];
    rq.plane(3)[i] = r.dir[0];
    rq.plane(4)[i] = r.dir[1];
    rq.plane(5)[i] = r.dir[2];
    rq.plane(6)[i] = r.tmin;
    rq.plane(7)[i] = r.tmax;
    rq.ids()[i] = k;
}
__global__ void k_hits_soa_to_aos(HitQ hq, uint32_t n, uint32_t per, uint32_t region, uint32_t mode, skh_hit* __restrict__ hits,
                                  const skh_instance* __restrict__ shadeInst /* light_id of a mesh instance = its first shading record */)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const uint32_t g = k / per, i = g * region + (k - g * per);
    skh_hit h;
    if (mode == SKH_TRACE_SHADOW)
    {
        h.t = hq.base[i];
        h.instance_id = h.prim_id = 0xffffffffu;
        h.u = h.v = 0.0f;
    }
    else
    {
        const float4 r0 = hq.rec(i)[0], r1 = hq.rec(i)[1];
        h.t = r0.x, h.u = r0.y, h.v = r0.z;
        h.instance_id = __float_as_uint(r1.x);
        h.prim_id = __float_as_uint(r1.y);
        // a hit on a baked triangle carries the index of its shading record (k_gather_tris): back to the primitive index inside the mesh
        if (h.instance_id != 0xffffffffu && (h.prim_id & SKH_PRIM_DIRECT))
            h.prim_id = (h.prim_id & ~SKH_PRIM_DIRECT) - shadeInst[h.instance_id].light_id;
    }
    hits[k] = h;
}

// ---- memory ceilings of this GPU, measured with the access shapes of the hot path (include/strelka_hip.h: skh_probe_memory) ----
__global__ __launch_bounds__(256) void k_probe_fill(uint4* __restrict__ buf, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const uint32_t h = hash_murmur((uint32_t)i * 0x9e3779b9u + (uint32_t)(i >> 32));
        buf[i] = make_uint4(h, h * 0x85ebca6bu, h ^ 0xc2b2ae35u, h * 0x27d4eb2fu + 1u);
    }
}

__global__ __launch_bounds__(256) void k_probe_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// one-wave workgroups like k_trace; every lane fetches `perLane` records of R uint4 (R = 4: 64 bytes = one BVH node) at pseudo-random,
// record-aligned places of the buffer.  DEPENDENT: the next place comes out of the record just loaded (a traversal step); otherwise
// four fetches are in flight.
template <bool DEPENDENT, int R>
__global__ __launch_bounds__(SKH_TRACE_BLOCK) void k_probe_gather(const uint4* __restrict__ buf, uint32_t nRec, uint32_t perLane, uint32_t* __restrict__ sink)
{
    uint32_t s = hash_murmur(blockIdx.x * SKH_TRACE_BLOCK + threadIdx.x + 1u);
    uint32_t acc = 0u;
    if (DEPENDENT)
    {
        for (uint32_t i = 0; i < perLane; ++i)
        {
            const uint4* p = buf + R * (size_t)__umulhi(s, nRec);
            uint32_t x = 0u;
#pragma unroll
            for (int j = 0; j < R; ++j)
            {
                const uint4 a = p[j];
                x ^= a.x ^ a.y ^ a.z ^ a.w;
            }
            acc += x;
            s = s * 1664525u + 1013904223u + x;
        }
    }
    else
    {
        for (uint32_t i = 0; i < perLane; i += 4)
        {
            uint4 v[4][R];
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint4* p = buf + R * (size_t)__umulhi(s, nRec);
                s = s * 1664525u + 1013904223u;
#pragma unroll
                for (int j = 0; j < R; ++j)
                    v[k][j] = p[j];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < R; ++j)
                    acc += v[k][j].x ^ v[k][j].y ^ v[k][j].z ^ v[k][j].w;
        }
    }
    if (acc == 0x12345u) // (keeps the loads alive; practically never true)
        sink[0] = s;
}

skh_status skh_probe_memory(skh_context* c, uint32_t kind, uint64_t bytes, uint32_t record_bytes, uint32_t repeat, double* out_gbps, double* out_ms)
{
    if (!c || kind > SKH_PROBE_CHASE || bytes < (1u << 20) || !out_gbps ||
        (kind != SKH_PROBE_COPY && record_bytes != 32 && record_bytes != 64 && record_bytes != 128))
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    DevBuf a, b;
    skh_status s;
    const size_t n16 = (size_t)(bytes / 64) * 4; // uint4 elements, whole 64-byte records
    if ((s = dev_alloc(c, a, n16 * 16)) != SKH_OK || (kind == SKH_PROBE_COPY && (s = dev_alloc(c, b, n16 * 16)) != SKH_OK))
    {
        dev_free(a);
        dev_free(b);
        return s;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        if (e0)
            (void)hipEventDestroy(e0);
        if (e1)
            (void)hipEventDestroy(e1);
        dev_free(a);
        dev_free(b);
    };
    const uint32_t grid = (uint32_t)c->numCUs * c->wavesPerCU; // the trace kernels' grid
    const uint32_t perLane = 256;
    repeat = std::max(1u, repeat);
    k_probe_fill<<<c->numCUs * 8, 256, 0, c->stream>>>(a.as<uint4>(), n16);
    double moved = 0.0;
    for (uint32_t r = 0; r <= repeat; ++r) // (pass 0 is the warm-up)
    {
        if (r == 1)
        {
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventRecord(e0, c->stream) != hipSuccess)
            {
                cleanup();
                c->err = "skh_probe_memory: hipEvent";
                return SKH_FAIL;
            }
        }
        if (kind == SKH_PROBE_COPY)
        {
            k_probe_copy<<<c->numCUs * 16, 256, 0, c->stream>>>(a.as<uint4>(), b.as<uint4>(), n16);
            moved = 2.0 * 16.0 * (double)n16;
        }
        else
        {
            const uint32_t nRec = (uint32_t)(n16 * 16 / record_bytes);
            uint32_t* sink = a.as<uint32_t>();
#define SKH_PROBE_LAUNCH(DEP, R) k_probe_gather<DEP, R><<<grid, SKH_TRACE_BLOCK, 0, c->stream>>>(a.as<uint4>(), nRec, perLane, sink)
            if (kind == SKH_PROBE_CHASE)
            {
                if (record_bytes == 32)
                    SKH_PROBE_LAUNCH(true, 2);
                else if (record_bytes == 64)
                    SKH_PROBE_LAUNCH(true, 4);
                else
                    SKH_PROBE_LAUNCH(true, 8);
            }
            else
            {
                if (record_bytes == 32)
                    SKH_PROBE_LAUNCH(false, 2);
                else if (record_bytes == 64)
                    SKH_PROBE_LAUNCH(false, 4);
                else
                    SKH_PROBE_LAUNCH(false, 8);
            }
#undef SKH_PROBE_LAUNCH
            moved = (double)record_bytes * perLane * SKH_TRACE_BLOCK * (double)grid;
        }
    }
    float ms = 0.0f;
    if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
    {
        cleanup();
        c->err = "skh_probe_memory: timing failed";
        return SKH_FAIL;
    }
    cleanup();
    ms /= (float)repeat;
    *out_gbps = moved / (ms * 1e-3) / 1e9;
    if (out_ms)
        *out_ms = ms;
    return SKH_OK;
}

skh_status skh_trace_device(skh_context* c, const void* d_rays, uint32_t n_rays, uint32_t mode, void* d_hits, uint32_t repeat)
{
    if (!c || (n_rays && (!d_rays || !d_hits)) || mode > 1)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    skh_status s = ensure_ready(c);
    if (s != SKH_OK || n_rays == 0)
        return s;
    // a speculative pass in flight shares the stack-overflow flag and the overflow area with this query: wait for it and drop it first, so
    // that an overflow is attributed to the call that caused it
    spec_drop(c, true);
    DevBuf q, h, cnt;
    auto cleanup = [&]() {
        dev_free(q);
        dev_free(h);
        dev_free(cnt);
    };
    const uint32_t per = (((n_rays + SKH_SHARDS - 1u) / SKH_SHARDS) + 63u) & ~63u; // rays per shard = the shard's region
    const size_t NQ = (size_t)SKH_SHARDS * per;
    if ((s = dev_alloc(c, q, sizeof(float) * 9 * NQ)) != SKH_OK || (s = dev_alloc(c, h, sizeof(float) * 8 * NQ)) != SKH_OK ||
        (s = dev_alloc(c, cnt, sizeof(uint32_t) * (SKH_SHARDS * SKH_COUNT_STRIDE + 8 * SKH_FETCH_STRIDE))) != SKH_OK)
    {
        cleanup();
        return s;
    }
    if (!c->traceBlocks)
        c->traceBlocks = (uint32_t)c->numCUs * c->wavesPerCU;
    if ((s = dev_alloc(c, c->dOvf, sizeof(int) * (size_t)(SKH_STACK_OVF + SKH_TAIL_EXTRA) * (uint32_t)c->numCUs * std::max(std::max(c->wavesPerCU, c->wavesPerCUWorld), std::max(c->wavesPerCUShadow, c->wavesPerCUShadowWorld)) * SKH_TRACE_BLOCK)) != SKH_OK)
    {
        cleanup();
        return s;
    }
    RayQ rq{ q.as<float>(), (uint32_t)NQ, per };
    HitQ hq{ h.as<float>(), (uint32_t)NQ };
    PathS ps{ nullptr, 0 };
    uint32_t* dcount = cnt.as<uint32_t>();
    uint32_t* dfetch = dcount + SKH_SHARDS * SKH_COUNT_STRIDE;
    uint32_t hcount[SKH_SHARDS * SKH_COUNT_STRIDE] = { 0 };
    for (uint32_t g = 0; g < SKH_SHARDS; ++g)
        hcount[g * SKH_COUNT_STRIDE] = std::min(per, n_rays - std::min(n_rays, g * per));
    SKH_TRY(c, hipMemcpyAsync(dcount, hcount, sizeof(hcount), hipMemcpyHostToDevice, c->stream));
    SKH_TRY(c, hipStreamSynchronize(c->stream)); // (hcount lives on this stack frame)
    const DevScene sc = make_dev_scene(c);
    k_rays_aos_to_soa<<<(n_rays + 255) / 256, 256, 0, c->stream>>>(reinterpret_cast<const skh_ray*>(d_rays), n_rays, per, rq);
    for (uint32_t r = 0; r < std::max(1u, repeat); ++r)
    {
        (void)hipMemsetAsync(dfetch, 0, sizeof(uint32_t) * 8 * SKH_FETCH_STRIDE, c->stream);
        SpanGuard g(c, mode == SKH_TRACE_SHADOW ? KC_TRACE_SHADOW : KC_TRACE_CLOSEST);
        if (mode == SKH_TRACE_SHADOW)
        {
            if (c->countTraversal)
                launch_trace<true, true>(c, sc, rq, dcount, dfetch, hq, ps, nullptr);
            else
                launch_trace<true, false>(c, sc, rq, dcount, dfetch, hq, ps, nullptr);
        }
        else
        {
            if (c->countTraversal)
                launch_trace<false, true>(c, sc, rq, dcount, dfetch, hq, ps, nullptr);
            else
                launch_trace<false, false>(c, sc, rq, dcount, dfetch, hq, ps, nullptr);
        }
    }
    k_hits_soa_to_aos<<<(n_rays + 255) / 256, 256, 0, c->stream>>>(hq, n_rays, per, per, mode, reinterpret_cast<skh_hit*>(d_hits), c->dShadeInst.as<skh_instance>());
    hipError_t e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess)
    {
        c->err = std::string("skh_trace: ") + hipGetErrorString(e);
        return SKH_FAIL;
    }
    if (c->timing)
        harvest_spans(c);
    return check_stack_overflow(c, "skh_trace");
}

skh_status skh_trace(skh_context* c, const skh_ray* rays, uint32_t n_rays, uint32_t mode, skh_hit* hits)
{
    if (!c || (n_rays && (!rays || !hits)))
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (n_rays == 0)
        return SKH_OK;
    DevBuf dr, dh;
    skh_status s;
    if ((s = dev_upload(c, dr, rays, sizeof(skh_ray) * (size_t)n_rays)) != SKH_OK || (s = dev_alloc(c, dh, sizeof(skh_hit) * (size_t)n_rays)) != SKH_OK)
    {
        dev_free(dr);
        dev_free(dh);
        return s;
    }
    s = skh_trace_device(c, dr.p, n_rays, mode, dh.p, 1);
    if (s == SKH_OK)
    {
        hipError_t e = hipMemcpy(hits, dh.p, sizeof(skh_hit) * (size_t)n_rays, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
        {
            c->err = std::string("skh_trace readback: ") + hipGetErrorString(e);
            s = SKH_FAIL;
        }
    }
    dev_free(dr);
    dev_free(dh);
    return s;
}

// ---- BSDF probes (tests) ----
static_assert(sizeof(skh_bsdf_query) == 84 && sizeof(skh_bsdf_result) == 64, "ABI layout");
__global__ void k_bsdf_probe(const skh_bsdf_query* __restrict__ q, uint32_t n, const Material* __restrict__ mats, uint32_t numMaterials,
                             skh_bsdf_result* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const skh_bsdf_query in = q[i];
    const Material m = mats[in.material < numMaterials ? in.material : 0u];
    const v3 N = mk3(in.normal[0], in.normal[1], in.normal[2]), Ng = mk3(in.geom_normal[0], in.geom_normal[1], in.geom_normal[2]);
    const v3 T = mk3(in.tangent_u[0], in.tangent_u[1], in.tangent_u[2]);
    const v3 k1 = mk3(in.k1[0], in.k1[1], in.k1[2]), k2 = mk3(in.k2[0], in.k2[1], in.k2[2]);
    const HairConst hc = hair_const(m); // (what k_hair_consts tabulates per material for k_shade)
    BsdfSample bs;
    bsdf_sample<true>(m, N, Ng, T, k1, in.xi[0], in.xi[1], in.xi[2], in.xi[3], in.inside != 0u, bs, &hc);
    BsdfEval ev;
    bsdf_evaluate<true>(m, N, Ng, T, k1, k2, in.inside != 0u, ev, &hc);
    skh_bsdf_result r;
    r.k2[0] = bs.k2.x, r.k2[1] = bs.k2.y, r.k2[2] = bs.k2.z;
    r.bsdf_over_pdf[0] = bs.bsdf_over_pdf.x, r.bsdf_over_pdf[1] = bs.bsdf_over_pdf.y, r.bsdf_over_pdf[2] = bs.bsdf_over_pdf.z;
    r.pdf = bs.pdf;
    r.event_type = bs.event_type;
    r.bsdf_diffuse[0] = ev.bsdf_diffuse.x, r.bsdf_diffuse[1] = ev.bsdf_diffuse.y, r.bsdf_diffuse[2] = ev.bsdf_diffuse.z;
    r.bsdf_glossy[0] = ev.bsdf_glossy.x, r.bsdf_glossy[1] = ev.bsdf_glossy.y, r.bsdf_glossy[2] = ev.bsdf_glossy.z;
    r.eval_pdf = ev.pdf;
    r.reserved0 = 0u;
    out[i] = r;
}

skh_status skh_bsdf_probe(skh_context* c, const skh_bsdf_query* queries, uint32_t n, skh_bsdf_result* results)
{
    if (!c || (n && (!queries || !results)))
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (n == 0)
        return SKH_OK;
    if (c->nMaterials == 0)
    {
        c->err = "skh_bsdf_probe: call skh_set_materials first";
        return SKH_INVALID_ARGUMENT;
    }
    DevBuf dq, dr;
    skh_status s;
    if ((s = dev_upload(c, dq, queries, sizeof(skh_bsdf_query) * (size_t)n)) != SKH_OK || (s = dev_alloc(c, dr, sizeof(skh_bsdf_result) * (size_t)n)) != SKH_OK)
    {
        dev_free(dq);
        dev_free(dr);
        return s;
    }
    k_bsdf_probe<<<(n + 127) / 128, 128, 0, c->stream>>>(dq.as<skh_bsdf_query>(), n, c->dMaterials.as<Material>(), c->nMaterials, dr.as<skh_bsdf_result>());
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)
        e = hipMemcpy(results, dr.p, sizeof(skh_bsdf_result) * (size_t)n, hipMemcpyDeviceToHost);
    dev_free(dq);
    dev_free(dr);
    if (e != hipSuccess)
    {
        c->err = std::string("skh_bsdf_probe: ") + hipGetErrorString(e);
        return SKH_FAIL;
    }
    return SKH_OK;
}

// ---- unit probes (tests): the device functions of the sampler, the light samplers and the accumulator, one call per record, so that
// GPU tests can hold the HIP code against the reference-generated fixtures of tests/golden/ directly ----
static const uint32_t kUnitIn[SKH_UNIT_COUNT] = { 20, 8, 20, 24, 12, 8, 12, 12, 8 }, kUnitOut[SKH_UNIT_COUNT] = { 12, 4, 48, 4, 16, 4, 12, 24, 40 };
static const uint32_t kUnitConst[SKH_UNIT_COUNT] = { 0, 0, 112, 112, 112, 0, 12, 12, 0 };
__global__ void __launch_bounds__(256) k_unit_probe(uint32_t unit, uint32_t param, const float* __restrict__ consts, const uint32_t* __restrict__ in, uint32_t n,
                                                    uint32_t* __restrict__ out)
{
    __shared__ uint32_t s_lut[SKH_SOBOL_LUT_WORDS]; // the byte-folded Sobol table as k_shade stages it
    for (uint32_t k = threadIdx.x; k < SKH_SOBOL_LUT_WORDS; k += blockDim.x)
        s_lut[k] = g_sobol_lut[k];
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float* fin = reinterpret_cast<const float*>(in);
    float* fout = reinterpret_cast<float*>(out);
    Light l;
    if (unit == SKH_UNIT_LIGHT_SAMPLE || unit == SKH_UNIT_LIGHT_PDF || unit == SKH_UNIT_LIGHT_NORMAL)
        l = *reinterpret_cast<const Light*>(consts);
    switch (unit)
    {
    case SKH_UNIT_SAMPLER: {
        const uint32_t* r = in + 5 * (size_t)i;
        Sampler s = init_sampler(r[0], r[1], r[2], param, 52u); // seed 52: OptixRender.cu:101
        s.depth = r[3];
        out[3 * (size_t)i] = __float_as_uint(sampler_random(s, r[4]));
        out[3 * (size_t)i + 1] = __float_as_uint(sampler_random_lut(s, r[4], s_lut));
        out[3 * (size_t)i + 2] = s.sampleIdx;
        break;
    }
    case SKH_UNIT_SOBOL:
        out[i] = sobol_uint(in[2 * (size_t)i], in[2 * (size_t)i + 1]);
        break;
    case SKH_UNIT_LIGHT_SAMPLE: {
        const float* r = fin + 5 * (size_t)i;
        const v3 P = mk3(r[0], r[1], r[2]);
        LightSample d;
        if (param == 0)
            d = sample_rect_light_uniform(l, r[3], r[4], P);
        else if (param == 1)
            d = sample_rect_light(l, r[3], r[4], P);
        else if (param == 2)
            d = sample_sphere_light(l, r[3], r[4], P);
        else
            d = sample_distant_light(l, r[3], r[4]);
        float* o = fout + 12 * (size_t)i;
        o[0] = d.pointOnLight.x, o[1] = d.pointOnLight.y, o[2] = d.pointOnLight.z, o[3] = d.pdf;
        o[4] = d.normal.x, o[5] = d.normal.y, o[6] = d.normal.z, o[7] = d.area;
        o[8] = d.L.x, o[9] = d.L.y, o[10] = d.L.z, o[11] = d.distToLight;
        break;
    }
    case SKH_UNIT_LIGHT_PDF: {
        const float* r = fin + 6 * (size_t)i;
        fout[i] = get_light_pdf(l, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]));
        break;
    }
    case SKH_UNIT_LIGHT_NORMAL: {
        const float* r = fin + 3 * (size_t)i;
        const v3 nn = calc_light_normal(l, mk3(r[0], r[1], r[2]));
        fout[4 * (size_t)i] = nn.x, fout[4 * (size_t)i + 1] = nn.y, fout[4 * (size_t)i + 2] = nn.z, fout[4 * (size_t)i + 3] = calc_light_area(l);
        break;
    }
    case SKH_UNIT_MIS:
        fout[i] = mis_weight_balance(fin[2 * (size_t)i], fin[2 * (size_t)i + 1]);
        break;
    case SKH_UNIT_ACCUMULATE: {
        // a SEQUENCE: record k is folded into the running value of records 0..k-1 (sub-frame index param + k): one thread
        if (i != 0)
            return;
        const v3 e = mk3(consts[0], consts[1], consts[2]);
        v3 prev = mk3(0.0f);
        for (uint32_t k = 0; k < n; ++k)
        {
            prev = accumulate(prev, mk3(fin[3 * (size_t)k], fin[3 * (size_t)k + 1], fin[3 * (size_t)k + 2]), e, param + k);
            fout[3 * (size_t)k] = prev.x, fout[3 * (size_t)k + 1] = prev.y, fout[3 * (size_t)k + 2] = prev.z;
        }
        break;
    }
    case SKH_UNIT_TONEMAP: {
        const v3 e = mk3(consts[0], consts[1], consts[2]);
        const v3 c = mk3(fin[3 * (size_t)i], fin[3 * (size_t)i + 1], fin[3 * (size_t)i + 2]);
        const v3 t = tonemap(c, e), iv = inverse_tonemap(c, e);
        float* o = fout + 6 * (size_t)i;
        o[0] = t.x, o[1] = t.y, o[2] = t.z, o[3] = iv.x, o[4] = iv.y, o[5] = iv.z;
        break;
    }
    case SKH_UNIT_LIBM: {
        // skh_libm.h on the device: the bits the CPU checker's copy of the same text must reproduce
        const float x = fin[2 * (size_t)i], y = fin[2 * (size_t)i + 1];
        float* o = fout + 10 * (size_t)i;
        o[0] = skm::sinf_(x), o[1] = skm::cosf_(x), o[2] = skm::acosf_(x), o[3] = skm::asinf_(x), o[4] = skm::atan2f_(y, x);
        o[5] = skm::expf_(x), o[6] = skm::logf_(x), o[7] = skm::sinhf_(x), o[8] = skm::powf_(x, y), o[9] = skm::atan2f_(x, y);
        break;
    }
    default:
        break;
    }
}

skh_status skh_unit_probe(skh_context* c, uint32_t unit, uint32_t param, const void* consts, const void* in, uint32_t n, void* out)
{
    if (!c || unit >= SKH_UNIT_COUNT || (n && (!in || !out)) || (kUnitConst[unit] && !consts))
    {
        if (c)
            c->err = "skh_unit_probe: unknown unit, or a missing input / output / constants pointer";
        return SKH_INVALID_ARGUMENT;
    }
    (void)hipSetDevice(c->device);
    if (n == 0)
        return SKH_OK;
    skh_status s; // (the Sobol tables were uploaded by skh_create)
    DevBuf di, dc, dout;
    if ((s = dev_upload(c, di, in, (size_t)kUnitIn[unit] * n)) != SKH_OK || (s = dev_alloc(c, dout, (size_t)kUnitOut[unit] * n)) != SKH_OK ||
        (kUnitConst[unit] && (s = dev_upload(c, dc, consts, kUnitConst[unit])) != SKH_OK))
    {
        dev_free(di), dev_free(dc), dev_free(dout);
        return s;
    }
    k_unit_probe<<<(n + 255) / 256, 256, 0, c->stream>>>(unit, param, dc.as<float>(), di.as<uint32_t>(), n, dout.as<uint32_t>());
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)
        e = hipMemcpy(out, dout.p, (size_t)kUnitOut[unit] * n, hipMemcpyDeviceToHost);
    dev_free(di), dev_free(dc), dev_free(dout);
    if (e != hipSuccess)
    {
        c->err = std::string("skh_unit_probe: ") + hipGetErrorString(e);
        return SKH_FAIL;
    }
    return SKH_OK;
}

skh_status skh_set_option(skh_context* c, const char* name, int64_t value)
{
    if (!c || !name)
        return SKH_INVALID_ARGUMENT;
    const std::string n(name);
    if (n != "timing" && n != "count_traversal")
        spec_drop(c);
    else if ((n == "count_traversal" && c->countTraversal != (value != 0)) || (n == "timing" && c->timing != (value != 0)))
        // a pass traced ahead with the OTHER setting must not be delivered under this one: its traversal counters would be missing from the
        // per-ray figures while its rays count; the pass in flight is waited for and dropped too
        spec_drop(c, true);
    if (n == "count_traversal")
        c->countTraversal = value != 0;
    else if (n == "timing")
        c->timing = value != 0;
    else if (n == "fetch_min_closest" || n == "fetch_min_shadow")
    {
        if (value < 1 || value > 64)
            return SKH_INVALID_ARGUMENT;
        if (n == "fetch_min_closest")
            c->fetchMinClosestSet = true;
        (n == "fetch_min_closest" ? c->fetchMinClosest : c->fetchMinShadow) = (uint32_t)value; // (an explicit value applies to both builds)
        (n == "fetch_min_closest" ? c->curveFetchMinClosest : c->curveFetchMinShadow) = (uint32_t)value;
    }
    else if (n == "fetch_min_closest_small")
    {
        if (value < 1 || value > 64)
            return SKH_INVALID_ARGUMENT;
        c->fetchMinClosestSmall = (uint32_t)value;
    }
    else if (n == "compact_hits")
    {
        if (value != 0 && value != 1)
            return SKH_INVALID_ARGUMENT;
        c->compactHits = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false; // (skh_build_accel decides with it which word a baked triangle's hit carries)
    }
    else if (n == "direct_records")
    {
        if (value < -1 || value > 1)
            return SKH_INVALID_ARGUMENT;
        c->directRecordsOpt = (int32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "merge_light_proxies")
    {
        if (value != 0 && value != 1)
            return SKH_INVALID_ARGUMENT;
        c->mergeLightProxies = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "fetch_chunk")
    {
        if (value < -1 || value > 4096)
            return SKH_INVALID_ARGUMENT;
        c->fetchChunk = (int32_t)value;
    }
    else if (n == "curve_min")
    {
        if (value < 1 || value > 64)
            return SKH_INVALID_ARGUMENT;
        c->curveMin = c->worldCurveMin = (uint32_t)value;
    }
    else if (n == "leaf_min")
    {
        if (value < 0 || value > 64)
            return SKH_INVALID_ARGUMENT;
        c->leafMin = (uint32_t)value;
    }
    else if (n == "overlap")
    {
        if (value < 0 || value > 2)
            return SKH_INVALID_ARGUMENT;
        c->overlap = (int)value;
    }
    else if (n == "tight_instance_boxes")
    {
        c->tightInstanceBoxes = value != 0;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "curve_leaf")
    {
        if (value < 1 || value > 4)
            return SKH_INVALID_ARGUMENT;
        c->curveLeaf = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "split_pairs")
    {
        if (value < 0 || value > 1000)
            return SKH_INVALID_ARGUMENT;
        c->splitPairs = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "curve_merge")
    {
        if (value < 0 || value > 1)
            return SKH_INVALID_ARGUMENT;
        c->curveMerge = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "curve_segnode" || n == "curve_strand_major")
    {
        if (value < 0 || value > 1 || (value == 1 && !SKH_SEGNODE)) // (the kernels of this library were compiled without segment-node support: skh_kernels.h SKH_SEGNODE)
            return SKH_INVALID_ARGUMENT;
        (n == "curve_segnode" ? c->curveSegNode : c->curveStrandMajor) = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "curve_split")
    {
        if (value < 1 || value > 8)
            return SKH_INVALID_ARGUMENT;
        c->curveSplit = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "speculate")
    {
        if (value < 0 || value > 64)
            return SKH_INVALID_ARGUMENT;
        c->speculateMax = (uint32_t)value;
    }
    else if (n == "tlas_build")
    {
        if (value < 0 || value > 2)
            return SKH_INVALID_ARGUMENT;
        c->tlasBuild = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "tlas_open")
    {
        if (value < 0 || value > 64)
            return SKH_INVALID_ARGUMENT;
        c->tlasOpen = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "node_break_closest" || n == "node_break_shadow")
    {
        if (value < 0 || value > 64)
            return SKH_INVALID_ARGUMENT;
        // (an explicitly set value applies to the curve build too: its own defaults -- 20 / 20 -- hold until then)
        (n == "node_break_closest" ? c->nodeBreakClosest : c->nodeBreakShadow) = (uint32_t)value;
        (n == "node_break_closest" ? c->curveNodeBreakClosest : c->curveNodeBreakShadow) = (uint32_t)value;
    }
    else if (n == "subframe_batch")
    {
        if (value < 0 || value > 64)
            return SKH_INVALID_ARGUMENT;
        c->subframeBatch = (uint32_t)value;
        if (c->width)
            return alloc_frame(c);
    }
    else if (n == "bake_world")
    {
        if (value < 0 || value > 4)
            return SKH_INVALID_ARGUMENT;
        c->bakeWorld = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "world_kernel")
        c->worldKernel = value != 0;
    else if (n == "bake_budget_mtris")
    {
        if (value < 0 || value > 100)
            return SKH_INVALID_ARGUMENT; // (leaf references address 2^27 baked triangles)
        c->bakeBudgetMTris = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "bake_small_tris")
    {
        if (value < 0 || value > (1 << 20))
            return SKH_INVALID_ARGUMENT;
        c->bakeSmallTris = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "speculate_grow")
    {
        if (value < 2 || value > 64)
            return SKH_INVALID_ARGUMENT;
        c->speculateGrow = (uint32_t)value;
    }
    else if (n == "speculate_async")
    {
        if (value < 0 || value > 1)
            return SKH_INVALID_ARGUMENT;
        spec_drop(c);
        c->speculateAsync = (uint32_t)value;
    }
    else if (n == "morton_bits")
    {
        if (value < 4 || value > 21)
            return SKH_INVALID_ARGUMENT;
        c->mortonBits = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "leaf_lines")
    {
        if (value < 0 || value > 1)
            return SKH_INVALID_ARGUMENT;
        c->leafLines = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "leaf_max_tris")
    {
        if (value < 1 || value > 8)
            return SKH_INVALID_ARGUMENT;
        c->leafMaxTris = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "tail_split")
    {
        if (value < -1 || value > 2)
            return SKH_INVALID_ARGUMENT;
        c->tailSplit = (int)value;
    }
    else if (n == "small_waves_first" || n == "small_waves_last")
    {
        if (value < 0 || value > 64)
            return SKH_INVALID_ARGUMENT;
        (n == "small_waves_first" ? c->smallWavesFirst : c->smallWavesLast) = (uint32_t)value;
    }
    else if (n == "small_waves_closest" || n == "small_waves_shadow")
    {
        if (value < 0 || value > 64)
            return SKH_INVALID_ARGUMENT;
        (n == "small_waves_closest" ? c->smallWavesClosest : c->smallWavesShadow) = (uint32_t)value;
    }
    else if (n == "build_quality")
    {
        c->buildQuality = value != 0;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "reinsert_rounds" || n == "reinsert_curve_rounds" || n == "reinsert_min_size")
    {
        if (value < 0 || value > (n == "reinsert_min_size" ? (1 << 24) : 64))
            return SKH_INVALID_ARGUMENT;
        (n == "reinsert_rounds" ? c->reinsertRounds : (n == "reinsert_curve_rounds" ? c->reinsertCurveRounds : c->reinsertMinSize)) = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "ploc_top")
    {
        if (value < 0)
            return SKH_INVALID_ARGUMENT;
        c->plocTop = (uint32_t)value;
        c->accelBuilt = false, c->refitReady = false;
    }
    else if (n == "waves_per_cu" || n == "waves_per_cu_shadow" || n == "waves_per_cu_world" || n == "waves_per_cu_shadow_world")
    {
        if (value < 1 || value > 32)
            return SKH_INVALID_ARGUMENT;
        (n == "waves_per_cu" ? c->wavesPerCU : (n == "waves_per_cu_shadow" ? c->wavesPerCUShadow : (n == "waves_per_cu_world" ? c->wavesPerCUWorld : c->wavesPerCUShadowWorld))) = (uint32_t)value;
        if (c->width)
            return alloc_frame(c);
    }
    else
    {
        c->err = "skh_set_option: unknown option " + n;
        return SKH_INVALID_ARGUMENT;
    }
    return SKH_OK;
}

skh_status skh_get_device_info(skh_context* c, skh_device_info* out)
{
    if (!c || !out)
        return SKH_INVALID_ARGUMENT;
    hipDeviceProp_t prop;
    SKH_TRY(c, hipGetDeviceProperties(&prop, c->device));
    memset(out, 0, sizeof(*out));
    out->compute_units = (uint32_t)prop.multiProcessorCount;
    out->simds_per_cu = 4;
    out->clock_khz = (uint32_t)prop.clockRate;
    out->memory_clock_khz = (uint32_t)prop.memoryClockRate;
    out->memory_bus_bits = (uint32_t)prop.memoryBusWidth;
    out->wavefront_size = (uint32_t)prop.warpSize;
    out->total_memory_bytes = (uint64_t)prop.totalGlobalMem;
    strncpy(out->name, prop.name, sizeof(out->name) - 1);
    return SKH_OK;
}

skh_status skh_get_build_info(skh_context* c, skh_build_info* out)
{
    if (!c || !out)
        return SKH_INVALID_ARGUMENT;
    *out = c->buildInfo;
    out->ms_build = c->msBuild;
    return SKH_OK;
}

skh_status skh_get_stats(skh_context* c, skh_stats* out)
{
    if (!c || !out)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    StatsDev sd;
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    SKH_TRY(c, hipMemcpy(&sd, c->dStats.p, sizeof(sd), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    // rays count when their sub-frame is DELIVERED: what was traced ahead and thrown away (spec_drop) is out, and so is what is still
    // waiting to be collected -- the rest of the current pass and the whole pass in flight (complete after the synchronize above)
    uint64_t outR = c->discardedRadiance, outS = c->discardedShadow;
    {
        const skh_context::Speculation& sp = c->spec;
        if (sp.valid && sp.consumed < sp.count && sp.count > 0)
        {
            outR += sp.passRadiance * (sp.count - sp.consumed) / sp.count;
            outS += sp.passShadow * (sp.count - sp.consumed) / sp.count;
        }
        if (sp.nextInFlight)
        {
            outR += sd.raysRadiance - std::min<uint64_t>(sd.raysRadiance, sp.statsMark[0]);
            outS += sd.raysShadow - std::min<uint64_t>(sd.raysShadow, sp.statsMark[1]);
        }
    }
    out->rays_radiance = sd.raysRadiance - std::min<uint64_t>(sd.raysRadiance, outR);
    out->rays_shadow = sd.raysShadow - std::min<uint64_t>(sd.raysShadow, outS);
    out->speculated_discarded = c->discardedSubframes;
    for (int k = 0; k < 2; ++k)
    {
        out->nodes_visited[k] = sd.nodes[k];
        out->prims_tested[k] = sd.prims[k];
        out->segs_tested[k] = sd.segs[k];
        out->instances_entered[k] = sd.insts[k];
    }
#ifdef SKH_LANE_PROFILE
    for (int k = 0; k < 2; ++k)
        fprintf(stderr, "[lane-profile] %s: rays %llu nodes %llu (TLAS %llu) tris %llu insts %llu | wave: nodeIt %llu triIt %llu instBlk %llu outer %llu refills %llu refilled %llu | tri passes with: fp64 fallback %llu, sign test passed %llu, division %llu | curve blocks %llu (instBlk = Newton runs in the curve builds)\n",
                k ? "shadow" : "closest", k ? sd.raysShadow : sd.raysRadiance, sd.nodes[k], sd.segs[k], sd.prims[k], sd.insts[k], sd.wave[k][0], sd.wave[k][1],
                sd.wave[k][2], sd.wave[k][3], sd.wave[k][4], sd.wave[k][5], sd.wave[k][6], sd.wave[k][8], sd.wave[k][7], sd.wave[k][9]);
    {
        double tot = 0;
        for (int k = 0; k < 8; ++k)
            tot += (double)sd.shade[k];
        fprintf(stderr, "[shade-cycles] load %.1f%% hit+material %.1f%% bsdf_sample %.1f%% light %.1f%% bsdf_eval %.1f%% rest-of-hit %.1f%% tail+state %.1f%% compaction+queues %.1f%%\n",
                100 * sd.shade[0] / tot, 100 * sd.shade[1] / tot, 100 * sd.shade[2] / tot, 100 * sd.shade[3] / tot, 100 * sd.shade[4] / tot, 100 * sd.shade[5] / tot,
                100 * sd.shade[6] / tot, 100 * sd.shade[7] / tot);
    }
    fprintf(stderr, "[slow-rays] %u rays took more than 1500 node steps\n", sd.slowCount);
    for (uint32_t k = 0; k < std::min(sd.slowCount, 16u); ++k)
        fprintf(stderr, "[slow-ray] steps %.0f tris %.0f insts %.0f %s o %.9g %.9g %.9g d %.9g %.9g %.9g tmin %g tmax %g\n", sd.slow[k][0], sd.slow[k][1], sd.slow[k][2],
                sd.slow[k][3] != 0.0f ? "shadow" : "closest", sd.slow[k][4], sd.slow[k][5], sd.slow[k][6], sd.slow[k][7], sd.slow[k][8], sd.slow[k][9], sd.slow[k][10], sd.slow[k][11]);
    for (int k = 0; k < 2; ++k)
    {
        unsigned long long tr = 0, tb = 0;
        for (int j = 0; j < 42; ++j)
            tr += sd.runHist[k][j], tb += sd.blockHist[k][j];
        if (tr)
        {
            fprintf(stderr, "[newton-steps] %s runs %llu:", k ? "shadow" : "closest", tr);
            for (int j = 0; j < 42; ++j)
                fprintf(stderr, " %d:%.2f%%", j, 100.0 * (double)sd.runHist[k][j] / (double)tr);
            fprintf(stderr, "\n[newton-block-max] %s rounds %llu:", k ? "shadow" : "closest", tb);
            for (int j = 0; j < 42; ++j)
                fprintf(stderr, " %d:%.2f%%", j, 100.0 * (double)sd.blockHist[k][j] / (double)tb);
            fprintf(stderr, "\n");
        }
    }
    for (int k = 0; k < 2; ++k)
        fprintf(stderr, "[lane-cycles] %s: refill %.3g curve-block %.3g node %.3g leaf %.3g pop %.3g write %.3g total %.3g | shader clock while the waves ran: %.1f MHz (wave-seconds %.4g)\n", k ? "shadow" : "closest", (double)sd.cyc[k][0], (double)sd.cyc[k][8],
                (double)sd.cyc[k][1], (double)sd.cyc[k][2], (double)sd.cyc[k][3], (double)sd.cyc[k][4], (double)sd.cyc[k][5],
                sd.cyc[k][7] ? (double)sd.cyc[k][6] / (double)sd.cyc[k][7] * 100.0 : 0.0, (double)sd.cyc[k][7] / 1e8);
#endif
#ifdef SKH_TAIL_PROFILE
    for (int k = 0; k < 2; ++k)
    {
        unsigned long long waves = 0;
        double liveSum = 0;
        for (int j = 0; j < 64; ++j)
            waves += sd.exitHist[k][j];
        for (int j = 0; j < 65; ++j)
            liveSum += (double)j * (double)sd.dryLive[k][j];
        if (!waves)
            continue;
        fprintf(stderr, "[launch-tail] %s: %llu waves; lanes alive at a wave's dry point: mean %.1f; after it a wave stays %.1f us and spends %.1f ray-us (= %.1f lanes alive on average)\n", k ? "shadow" : "closest",
                waves, liveSum / (double)waves, (double)sd.waveTicksAfterDry[k] / 100.0 / (double)waves, (double)sd.rayTicksAfterDry[k] / 100.0 / (double)waves,
                sd.waveTicksAfterDry[k] ? (double)sd.rayTicksAfterDry[k] / (double)sd.waveTicksAfterDry[k] : 0.0);
        fprintf(stderr, "[launch-tail]   16-us bins: from the launch's first wave -- waves finding the queue dry %%, waves leaving %% | from a wave's own dry point -- waves leaving %%\n");
        for (int j = 0; j < 64; ++j)
            if (sd.dryHist[k][j] || sd.exitHist[k][j] || sd.afterDryHist[k][j])
                fprintf(stderr, "[launch-tail]   %4d us: dry %5.1f%%  exit %5.1f%% | exit after dry %5.1f%%\n", 16 * j, 100.0 * (double)sd.dryHist[k][j] / (double)waves, 100.0 * (double)sd.exitHist[k][j] / (double)waves,
                        100.0 * (double)sd.afterDryHist[k][j] / (double)waves);
    }
#endif
    out->ms_trace_closest = c->msClass[KC_TRACE_CLOSEST];
    out->ms_trace_shadow = c->msClass[KC_TRACE_SHADOW];
    out->ms_shade = c->msClass[KC_SHADE];
    out->ms_raygen = c->msClass[KC_RAYGEN];
    out->ms_accumulate = c->msClass[KC_ACCUM];
    out->ms_sort = c->msClass[KC_SORT];
    out->ms_build = c->msBuild;
    out->launches_trace_closest = c->launches[KC_TRACE_CLOSEST];
    out->launches_trace_shadow = c->launches[KC_TRACE_SHADOW];
    out->launches_shade = c->launches[KC_SHADE];
    out->launches_other = c->launches[KC_RAYGEN] + c->launches[KC_ACCUM];
    out->stack_overflows = c->stackOverflows;
    return SKH_OK;
}

skh_status skh_reset_stats(skh_context* c)
{
    if (!c)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    spec_drop(c, true); // (what was traced ahead belongs to the counters being cleared)
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    SKH_TRY(c, hipMemset(c->dStats.p, 0, sizeof(StatsDev)));
    c->stackOverflows = 0;
    c->discardedRadiance = c->discardedShadow = 0;
    c->discardedSubframes = 0;
    for (int k = 0; k < KC_COUNT; ++k)
    {
        c->msClass[k] = 0.0;
        c->launches[k] = 0;
    }
    return SKH_OK;
}

skh_status skh_synchronize(skh_context* c)
{
    if (!c)
        return SKH_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    SKH_TRY(c, hipStreamSynchronize(c->stream));
    return SKH_OK;
}

void* skh_get_stream(skh_context* c)
{
    return c ? (void*)c->stream : nullptr;
}

