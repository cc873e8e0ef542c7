// SNAPSHOT (round 4, commit 7fd1c86; not built): k_trace with the five measured-negative experiments still threaded through it --
// SKH_POP_CULL, SKH_POSTPONE, SKH_PREFETCH2, SKH_PK_NODE, 8-wide nodes (W8), continuations (TAILQ) -- see experiments/README.md and docs/LOG.md round 4.
#ifndef SKH_POP_CULL
#define SKH_POP_CULL 0 // 1: pop-time culling in the world-only closest-hit build: LDS stack entries are 64-bit {reference, entry distance} (one
                       // ds_write_b64 / ds_read_b64 each, SKH_CULL_LDS of them per lane) and a popped entry whose box lies beyond the current best
                       // hit is dropped WITHOUT fetching its node; a lane whose pop was culled stays in the node loop, masked, and pops again in
                       // the next iteration (no inner loop).  docs/LOG.md, round 4
#endif
#ifndef SKH_CULL_LDS
#define SKH_CULL_LDS 11 // 8 B x 64 lanes x 11 = 5632 B per wave: 28 waves per CU still fit 160 KB
#endif
#ifndef SKH_POSTPONE
#define SKH_POSTPONE 0 // 1: "speculative traversal" (Aila & Laine 2009) in the world-only closest-hit build: a lane that reaches a leaf puts it aside
                       // and keeps descending; only its SECOND leaf makes it wait for the wave's triangle pass, which then tests both
#endif
#ifndef SKH_TRI_COOP
#define SKH_TRI_COOP 1 // 1: the triangle pass of the world-only builds is shared -- lanes that are NOT at a leaf take the second triangle of the
                       // two-triangle leaves (the owner's ray pulled with ds_bpermute, their own ray state parked in the free part of their LDS stack
                       // column meanwhile), so that one pass does what took two at 30 + 17 of 64 lanes
#endif
#ifndef SKH_PK_NODE
#define SKH_PK_NODE 0 // 1: the near / far plane distances of a 4-wide node as v_pk_fma_f32 pairs (12 packed FMAs instead of 24).  Measured in round 4:
                      // the register pairs cost the closest-hit build 5 spilled dwords -- kitchen 81.3 -> 87.3 ms, unshared 70.8 -> 77.2; off
#endif
#ifndef SKH_PREFETCH2
#define SKH_PREFETCH2 0 // 1: touch load of the second-nearest hit child's line, issued behind the nearest child's node fetch
#endif
// The 8 ray-fetch cursors of a launch sit in separate 128-byte lines: returning atomics on ONE line serialise at ~88 per
// microsecond chip-wide (measured), which eight cursors in the same line would share.
#define SKH_FETCH_STRIDE 32
#define SKH_COUNT_STRIDE 32 // same for the queue-length words the compaction atomics hit

// The tail of a closest-hit launch (round 4).  A persistent launch ends when its LAST ray ends: once the queue is dry the waves thin out and
// the launch waits ~0.3 ms for a few long rays at a handful of lanes per wave -- per bounce, before the dependent k_shade may start; a rank's
// 1/8 share of a frame loses 9 % to that.  Continuations take the per-bounce barrier away from those rays: a wave that finds the queue dry
// PARKS the rays it still carries (the ray, its best hit, current node and stack: a record in one of eight per-shard lists), marks their queue
// entries (high bit of the id word: k_shade skips them) and exits.  The NEXT closest-hit launch takes the parked rays first (`resume`), 64 to
// a wave again, among a full launch's worth of other work; their results go back into their records and the k_shade launch after it shades
// them too ("late" rays, read from the records) -- one launch later than their queue mates, which their id word records as a LAG (bits 28-30):
// the bounce index of a ray is `launch index - lag`, and what it emits inherits the lag.  A path may be parked `lagMax` times; that many extra
// launch rounds at the end of the pass drain the stragglers.  Paths are independent, a path has one ray in flight, and every sum it takes
// part in stays in its own bounce order: images are bit-identical (tests/test_gpu_parity.py::test_tail_passes_are_exact).
// Record = SKH_TAIL_HDR + SKH_STACK_LDS words, in planes of SKH_SHARDS * capacity.
#define SKH_TAIL_HDR 16 // id word (path | lag << 28) | cur | sp, found << 31 (0xffffffff: moved on to the next list) | best t u v inst prim | leaf put aside | o xyz d xyz tmin
#define SKH_PARKED_BIT 0x80000000u
#define SKH_LAG_SHIFT 28
#define SKH_PATH_MASK 0x0fffffffu
struct TailQ
{
    // one list per queue shard (a late ray is shaded into the output shard of ITS input shard, so that no shard can outgrow its region):
    // list g = records [g * capacity, g * capacity + min(count[g], capacity)); a list that is full takes no more -- those rays stay in their wave
    uint32_t* park; // records this launch parks
    uint32_t* parkCount; // SKH_SHARDS words, SKH_COUNT_STRIDE apart (own 128-byte lines), then SKH_SHARDS copies of the "queue is dry" flag, same spacing
    uint32_t* budget; // SKH_SHARDS words, same spacing: rays parked per shard in the whole PASS, capped at `capacity` -- so a shard never holds more
                      // than `capacity` lagging rays, and the drain rounds' k_shade grids can be sized by it
    uint32_t* resume; // records the launch before parked: taken first, results written back into them
    const uint32_t* resumeCount; // their lists' lengths
    uint32_t* resumeFetch; // SKH_SHARDS cursors, SKH_FETCH_STRIDE apart
    uint32_t capacity; // records per list
    uint32_t parkMax, lagMax;
    // (kernel argument beside the pointer to this struct -- flags: 1 = may park (when the queue is dry and at most `parkMax` lanes of the wave still
    // carry a ray), 2 = has parked rays to resume)
    __device__ static uint32_t* plane(uint32_t* base, uint32_t cap, uint32_t k)
    {
        return base + (size_t)k * (SKH_SHARDS * cap);
    }
    __device__ uint32_t* dry(uint32_t g) const // (a copy per workgroup label: thousands of waves poll it)
    {
        return parkCount + (SKH_SHARDS + g) * SKH_COUNT_STRIDE;
    }
};

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
// TAILQ: the build with the continuation code in it (TailQ: park / resume); the launches that never park run the build without it -- the
// extra paths cost the traversal loop registers (13 spilled dwords) even when they are never taken.
template <bool ANY_HIT, bool COUNT, bool CURVES, bool W8 = false, bool WORLD = false, bool TAILQ = false>
__global__ void __launch_bounds__(SKH_TRACE_BLOCK, WORLD ? (W8 || TAILQ ? SKH_WORLD_MIN_WAVES : (ANY_HIT ? SKH_WORLD_ANYHIT_MIN_WAVES : SKH_WORLD_CLOSEST_MIN_WAVES)) : (CURVES ? SKH_CURVE_MIN_WAVES : (ANY_HIT ? SKH_ANYHIT_MIN_WAVES : SKH_TRACE_MIN_WAVES))) SKH_TRACE_ATTR
    k_trace(DevScene sc, RayQ rq, const uint32_t* __restrict__ countPtr, uint32_t* __restrict__ fetch /*8 counters, zeroed*/,
            uint32_t fetchArg /* refill threshold | curve-test threshold << 8 | node-break threshold << 16 | leaf-kind threshold << 24 */, 
            HitQ hq, PathS ps, const float* __restrict__ contrib, uint32_t contribStride, int* __restrict__ ovfBase,
            StatsDev* __restrict__ stats, const TailQ* __restrict__ tailp /* this launch's continuation lists: read where they are used (rare paths), not
            held in SGPRs through the traversal loop */, uint32_t tailFlagsArg)
{
    static_assert(!WORLD || !CURVES, "the world-only build is a triangle kernel");
    constexpr bool CULL = SKH_POP_CULL && WORLD && !ANY_HIT && !W8;
    constexpr bool TRICOOP = SKH_TRI_COOP && WORLD && !W8 && !SKH_POSTPONE && !SKH_POP_CULL; // (closest-hit and any-hit builds of the world-only kernel)
    // per-lane stack entries in LDS (the rest: SKH_STACK_OVF entries in global memory); TRICOOP gives one up for its two 64-byte lane tables
    // (LDS is handed out in 1280-byte granules here: 20 x 256 B = 4 granules exactly, 128 B more would cost a fifth = 25 instead of 28 waves per CU)
    constexpr int NLDS = CULL ? SKH_CULL_LDS : (TRICOOP ? SKH_STACK_LDS - 1 : SKH_STACK_LDS);
    __shared__ int s_stack[(CULL ? 2 : 1) * NLDS * SKH_TRACE_BLOCK];
    __shared__ unsigned char s_tab[TRICOOP ? 128 : 1]; // [0..63] owner lanes by rank, [64..127] helper lanes by rank
    const uint32_t fetchMin = fetchArg & 0xffu, curveMin = (fetchArg >> 8) & 0xffu, nodeBreak = (fetchArg >> 16) & 0xffu, leafMin = fetchArg >> 24;
    const uint32_t lane = threadIdx.x;
    constexpr bool TAILS = TAILQ && WORLD && !ANY_HIT && !W8 && !SKH_POP_CULL; // the build that can park / resume rays (TailQ)
    const uint32_t tailFlags = TAILS ? tailFlagsArg : 0u;
#define tail (*tailp)
    uint32_t phase = (tailFlags & 2u) ? 0u : 1u; // where a refill looks: 0 = the parked rays of the launch before (first), 1 = the ray queue
    uint32_t n = 0; // (countPtr: SKH_SHARDS queue-length words, SKH_COUNT_STRIDE apart)
#pragma unroll
    for (uint32_t g = 0; g < SKH_SHARDS; ++g)
        n += countPtr[g * SKH_COUNT_STRIDE] + ((tailFlags & 2u) ? tail.resumeCount[g * SKH_COUNT_STRIDE] : 0u);
    if (n == 0)
        return;
    const uint32_t perGroup = rq.region;
    const uint32_t group = blockIdx.x & 7u;
    uint32_t tries = 0;
    bool exhausted = false;
    int* lds = s_stack + lane;
    int2* lds2 = reinterpret_cast<int2*>(s_stack) + lane; // (CULL) entry e of this lane = lds2[e * 64] = {reference, entry distance}
    // (the overflow area is addressed from ovfBase where it is used -- rare paths -- instead of through a per-lane 64-bit pointer held across the loops)
#define SKH_OVF_AT(e) ovfBase[(size_t)(e) * ovfStride + (blockIdx.x * SKH_TRACE_BLOCK + threadIdx.x)]
    const uint32_t ovfStride = gridDim.x * SKH_TRACE_BLOCK;
    const uint32_t rayMask = CURVES ? (ANY_HIT ? 3u : 255u) : (ANY_HIT ? 1u : 253u);
    TraceCounters tc = { 0, 0, 0, 0 };
#ifdef SKH_LANE_PROFILE
    uint32_t wv[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    uint32_t rayNodes = 0, rayTris = 0, rayInsts = 0;
    unsigned long long cy[6] = { 0, 0, 0, 0, 0, 0 };
    const unsigned long long cyStart = __builtin_readcyclecounter();
    const unsigned long long rtStart = __builtin_amdgcn_s_memrealtime(); // constant 100 MHz counter: cycles / realtime = the clock this launch really ran at
#define SKH_LP(...) __VA_ARGS__
#else
#define SKH_LP(...)
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
    constexpr bool POSTPONE = SKH_POSTPONE && WORLD && !ANY_HIT && !W8;
    int leaf2 = SKH_REF_INVALID; // (POSTPONE) the leaf this lane has put aside
    uint32_t dryNext = 0; // (TAILS) the "queue is dry" flag as loaded one iteration ago
    uint32_t pollTick = 0;
    bool parkTried = false; // (TAILS) a wave offers its rays to the tail lists once (a full list must not be hammered every iteration)
    constexpr bool PF2 = SKH_PREFETCH2 && WORLD && !ANY_HIT && !W8;
    int pf = SKH_REF_INVALID; // (PF2) the second-nearest hit child of the node just processed: its line is touched behind the next node fetch
    int pfv = 0;
    HitRec best;
    best.t = 0.0f, best.inst = best.prim = 0xffffffffu, best.u = best.v = 0.0f, best.found = false;

#define SKH_PUSH_T(v, tnearBits)                                     \
    {                                                                \
        if (sp < NLDS)                                               \
        {                                                            \
            if (CULL)                                                \
                lds2[sp * SKH_TRACE_BLOCK] = make_int2((v), (tnearBits)); \
            else                                                     \
                lds[sp * SKH_TRACE_BLOCK] = (v);                     \
        }                                                            \
        else if (sp < NLDS + SKH_STACK_OVF)                          \
            SKH_OVF_AT(sp - NLDS) = (v); /* (entries in the global overflow area carry no distance: never culled) */ \
        else                                                         \
            *sc.overflowFlag = 1u; /* the entry is dropped: the call that launched this kernel returns SKH_FAIL, never silent */ \
        ++sp;                                                        \
    }
#define SKH_PUSH(v) SKH_PUSH_T(v, 0)
#define SKH_POP(dst)                                                 \
    {                                                                \
        --sp;                                                        \
        if (sp < NLDS)                                               \
        {                                                            \
            if (CULL)                                                \
            {                                                        \
                /* pop-time culling: the acceptance test the entry passed when it was pushed, against today's best.t -- an entry is dropped   \
                   only if the slab test would reject it now, so results cannot change */                                                    \
                const int2 e = lds2[sp * SKH_TRACE_BLOCK];           \
                dst = __int_as_float(e.y) > best.t * SKH_SLAB_SLACK ? SKH_REF_INVALID : e.x; \
            }                                                        \
            else                                                     \
                dst = lds[sp * SKH_TRACE_BLOCK];                     \
        }                                                            \
        else if (sp < NLDS + SKH_STACK_OVF)                          \
        {                                                            \
            dst = SKH_OVF_AT(sp - NLDS);              \
            if (PF2)                                                 \
                asm volatile("" ::"v"(dst)); /* the wait for this (rare) global read stays inside its branch: at the join it would cover the touch load in flight too */ \
        }                                                            \
        else                                                         \
            dst = SKH_REF_INVALID;                                   \
    }

    for (;;)
    {
        // ---------------- refill idle lanes from the queue ----------------
        const unsigned long long needMask = __ballot(!hasRay);
        const uint32_t want = (uint32_t)__popcll(needMask);
        SKH_LP(wv[3]++; unsigned long long cyA = __builtin_readcyclecounter();)
        if (want >= fetchMin || want == 64u)
        {
            // results of the lanes that finished since the last refill: written together, once per refill
            if (pending)
            {
                pending = false;
                const uint32_t i = ridx;
                if (ANY_HIT)
                {
                    if (hq.base) // raw query mode (skh_trace): 1 = occluded, -1 = not
                        hq.base[i] = best.found ? 1.0f : -1.0f;
                    else if (!best.found)
                    {
                        const uint32_t pid = rq.ids()[i];
                        float* rad = ps.base + (size_t)3 * ps.stride;
#if SKH_SHADOW_ATOMIC
                        // a path has at most one shadow ray in a launch, so a fire-and-forget add gives the bits `+=` gives -- without the
                        // wave sitting through the load -> add -> store round trip at every refill
                        unsafeAtomicAdd(&rad[pid], contrib[i]);
                        unsafeAtomicAdd(&rad[pid + ps.stride], contrib[i + contribStride]);
                        unsafeAtomicAdd(&rad[pid + 2 * (size_t)ps.stride], contrib[i + 2 * (size_t)contribStride]);
#else
                        rad[pid] += contrib[i];
                        rad[pid + ps.stride] += contrib[i + contribStride];
                        rad[pid + 2 * (size_t)ps.stride] += contrib[i + 2 * (size_t)contribStride];
#endif
                    }
                }
                else if (TAILS && (i & 0x80000000u))
                {
                    // a resumed ray: its hit goes back into its record, where the late part of k_shade finds it
                    const uint32_t k = i & 0x7fffffffu;
                    TailQ::plane(tail.resume, tail.capacity, 2)[k] = best.found ? 0x80000000u : 0u;
                    TailQ::plane(tail.resume, tail.capacity, 3)[k] = __float_as_uint(best.t);
                    TailQ::plane(tail.resume, tail.capacity, 4)[k] = __float_as_uint(best.u);
                    TailQ::plane(tail.resume, tail.capacity, 5)[k] = __float_as_uint(best.v);
                    TailQ::plane(tail.resume, tail.capacity, 6)[k] = best.inst;
                    TailQ::plane(tail.resume, tail.capacity, 7)[k] = best.prim;
                }
                else
                {
                    float4* hr = hq.rec(i);
                    hr[0] = make_float4(best.found ? best.t : -1.0f, best.u, best.v, 0.0f);
                    hr[1] = make_float4(__uint_as_float(best.inst), __uint_as_float(best.prim), 0.0f, 0.0f);
                }
            }
        }
        if (!exhausted && (want >= fetchMin || want == 64u))
        {
            SKH_LP(wv[4]++; wv[5] += want;)
            uint32_t base = 0, count = 0;
            const int leader = __ffsll((long long)needMask) - 1;
            bool fromResume = false;
            for (;;)
            {
                // the work of this phase: the eight lists of parked rays (phase 0) or the eight shards of the queue (phase 1); same cursor logic
                const bool ph0 = TAILS && phase == 0u;
                const uint32_t* __restrict__ cntPtr = ph0 ? tail.resumeCount : countPtr;
                uint32_t* cursors = ph0 ? tail.resumeFetch : fetch;
                const uint32_t span = ph0 ? tail.capacity : perGroup;
                while (tries < 8u)
                {
                    const uint32_t g = (group + tries) & 7u;
                    uint32_t b = 0;
                    if ((int)lane == leader)
                        b = atomicAdd(&cursors[g * SKH_FETCH_STRIDE], want);
                    b = __shfl(b, leader);
                    const uint32_t lo = g * span;
                    const uint32_t hi = lo + min(cntPtr[g * SKH_COUNT_STRIDE], span);
                    if (lo < hi && b < hi - lo)
                    {
                        base = lo + b;
                        count = min(want, hi - base);
                        if (count < want)
                            ++tries; // this range is now empty
                        break;
                    }
                    ++tries;
                }
                fromResume = ph0;
                if (ph0 && tries >= 8u && count == 0)
                {
                    phase = 1u; // no parked rays left: on to the queue, in this same refill
                    tries = 0;
                    continue;
                }
                break;
            }
            if (tries >= 8u && count == 0)
            {
                exhausted = true;
                if (TAILS && (tailFlags & 1u) && lane < SKH_SHARDS)
                    __hip_atomic_store(tail.dry(lane), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // every ray has been handed out: tell the other waves
            }
            const uint32_t rank = rank_below(needMask);
            if (TAILS && fromResume)
            {
                // ---- resume: a parked ray's record instead of a queue entry ----
                if (!hasRay && rank < count)
                {
                    const uint32_t k = base + rank;
                    uint32_t* R = tail.resume;
                    const uint32_t C = tail.capacity;
                    ridx = k | 0x80000000u; // (bit 31: "late" -- the result goes back into record k)
                    cur = (int)TailQ::plane(R, C, 1)[k];
                    const uint32_t spw = TailQ::plane(R, C, 2)[k];
                    sp = (int)(spw & 0x7fffffffu);
                    best.found = (spw >> 31) != 0u;
                    best.t = __uint_as_float(TailQ::plane(R, C, 3)[k]), best.u = __uint_as_float(TailQ::plane(R, C, 4)[k]), best.v = __uint_as_float(TailQ::plane(R, C, 5)[k]);
                    best.inst = TailQ::plane(R, C, 6)[k], best.prim = TailQ::plane(R, C, 7)[k];
                    leaf2 = (int)TailQ::plane(R, C, 8)[k];
                    ow = mk3(__uint_as_float(TailQ::plane(R, C, 9)[k]), __uint_as_float(TailQ::plane(R, C, 10)[k]), __uint_as_float(TailQ::plane(R, C, 11)[k]));
                    dw = mk3(__uint_as_float(TailQ::plane(R, C, 12)[k]), __uint_as_float(TailQ::plane(R, C, 13)[k]), __uint_as_float(TailQ::plane(R, C, 14)[k]));
                    tmin = __uint_as_float(TailQ::plane(R, C, 15)[k]);
                    for (int e = 0; e < sp; ++e) // (parked with sp <= NLDS)
                        lds[e * SKH_TRACE_BLOCK] = (int)TailQ::plane(R, C, SKH_TAIL_HDR + e)[k];
                    o = ow;
                    d = dw;
                    inv = rcp3(d);
                    sh = make_shear(dw);
                    nodes = sc.triNodes;
                    inBlas = true;
                    curInst = 0xffffffffu;
                    curType = 0;
                    pend = 0;
                    hasRay = true;
                }
            }
            else if (!hasRay && rank < count)
            {
                ridx = base + rank;
                ow = mk3(rq.plane(0)[ridx], rq.plane(1)[ridx], rq.plane(2)[ridx]);
                dw = mk3(rq.plane(3)[ridx], rq.plane(4)[ridx], rq.plane(5)[ridx]);
                tmin = rq.plane(6)[ridx];
                o = ow;
                d = dw;
                inv = rcp3(d);
                if (ANY_HIT && !WORLD)
                    invw = inv;
                const int wr0 = sc.worldRoot, wr1 = ANY_HIT ? SKH_REF_INVALID : sc.lightRoot; // (kernel arguments: scalar branches)
                if (wr0 != SKH_REF_INVALID || wr1 != SKH_REF_INVALID)
                {
                    // baked instances first: the ray starts INSIDE their world-space groups (identity entry: o = ow, d = dw), the top
                    // level waits under a sentinel on the stack
                    sh = make_shear(dw);
                    nodes = sc.triNodes;
                    inBlas = true;
                    curInst = 0xffffffffu; // = "the instance id is in the triangle record"
                    curType = 0;
                    sp = 0;
                    if (!WORLD && sc.tlasRoot != SKH_REF_INVALID)
                    {
                        lds[0] = sc.tlasRoot;
                        lds[SKH_TRACE_BLOCK] = SKH_REF_SENTINEL;
                        sp = 2;
                    }
                    if (wr0 != SKH_REF_INVALID && wr1 != SKH_REF_INVALID)
                    {
                        if (CULL)
                            lds2[sp * SKH_TRACE_BLOCK] = make_int2(wr1, 0); // (no entry distance known: never culled)
                        else
                            lds[sp * SKH_TRACE_BLOCK] = wr1;
                        ++sp;
                    }
                    cur = wr0 != SKH_REF_INVALID ? wr0 : wr1;
                }
                else
                {
                    nodes = sc.tlasNodes;
                    inBlas = false;
                    sp = 0;
                    cur = sc.tlasRoot;
                }
                best.t = rq.plane(7)[ridx];
                best.inst = best.prim = 0xffffffffu;
                best.u = best.v = 0.0f;
                best.found = false;
                pend = 0;
                leaf2 = SKH_REF_INVALID;
                hasRay = true;
            }
        }
        if (TAILS && (tailFlags & 1u))
        {
            // a wave learns that the queue is dry when its own refill fails -- which it only attempts with `fetchMin` idle lanes -- or from the
            // flag the first such wave sets (lane 0 polls its label's copy every fourth iteration, one poll ahead: the load is never waited for)
            if (!exhausted && phase == 1u && (++pollTick & 3u) == 0u)
            {
                if (__builtin_amdgcn_readfirstlane((int)dryNext) != 0)
                    exhausted = true;
                if (lane == 0u)
                    dryNext = __hip_atomic_load(tail.dry(group), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // ---- park: the queue is dry and this wave is down to a few rays -- leave them to the next launch and go ----
            const unsigned long long live = __ballot(hasRay);
            if (exhausted && !parkTried && live != 0ull && (uint32_t)__popcll(live) <= tail.parkMax)
            {
                parkTried = true;
                const bool late = (ridx & 0x80000000u) != 0u;
                uint32_t idw = 0;
                if (hasRay) // the ray's id word: path | lag << 28
                    idw = late ? TailQ::plane(tail.resume, tail.capacity, 0)[ridx & 0x7fffffffu] : rq.ids()[ridx];
                const uint32_t lag = (idw >> SKH_LAG_SHIFT) & 7u;
                // (a ray with entries in the global overflow area stays: rare; so does a path that has used up its lag allowance)
                const bool canPark = hasRay && sp <= NLDS && lag < tail.lagMax;
                const uint32_t myShard = !canPark ? 0xffffffffu : (late ? (ridx & 0x7fffffffu) / tail.capacity : min(ridx / rq.region, SKH_SHARDS - 1u));
                bool parked = false;
                for (uint32_t g = 0; g < SKH_SHARDS; ++g)
                {
                    const unsigned long long m = __ballot(myShard == g);
                    if (m == 0ull)
                        continue;
                    const int leader = __ffsll((long long)m) - 1;
                    uint32_t b = 0xffffffffu;
                    if ((int)lane == leader)
                    {
                        const uint32_t cnt = (uint32_t)__popcll(m);
                        if (atomicAdd(&tail.budget[g * SKH_COUNT_STRIDE], cnt) + cnt <= tail.capacity) // (the pass's allowance for this shard)
                            b = atomicAdd(&tail.parkCount[g * SKH_COUNT_STRIDE], cnt);
                    }
                    b = __shfl(b, leader);
                    const uint32_t pos = b + rank_below(m);
                    if (myShard == g && b != 0xffffffffu) // (allowance used up: the rays stay in their wave)
                    {
                        const uint32_t k = g * tail.capacity + pos;
                        uint32_t* P = tail.park;
                        const uint32_t C = tail.capacity;
                        TailQ::plane(P, C, 0)[k] = (idw & SKH_PATH_MASK) | ((lag + 1u) << SKH_LAG_SHIFT); // shaded one launch later than it would have been
                        TailQ::plane(P, C, 1)[k] = (uint32_t)cur;
                        TailQ::plane(P, C, 2)[k] = (uint32_t)sp | (best.found ? 0x80000000u : 0u);
                        TailQ::plane(P, C, 3)[k] = __float_as_uint(best.t), TailQ::plane(P, C, 4)[k] = __float_as_uint(best.u), TailQ::plane(P, C, 5)[k] = __float_as_uint(best.v);
                        TailQ::plane(P, C, 6)[k] = best.inst, TailQ::plane(P, C, 7)[k] = best.prim;
                        TailQ::plane(P, C, 8)[k] = (uint32_t)leaf2;
                        TailQ::plane(P, C, 9)[k] = __float_as_uint(ow.x), TailQ::plane(P, C, 10)[k] = __float_as_uint(ow.y), TailQ::plane(P, C, 11)[k] = __float_as_uint(ow.z);
                        TailQ::plane(P, C, 12)[k] = __float_as_uint(dw.x), TailQ::plane(P, C, 13)[k] = __float_as_uint(dw.y), TailQ::plane(P, C, 14)[k] = __float_as_uint(dw.z);
                        TailQ::plane(P, C, 15)[k] = __float_as_uint(tmin);
                        for (int e = 0; e < sp; ++e)
                            TailQ::plane(P, C, SKH_TAIL_HDR + e)[k] = (uint32_t)lds[e * SKH_TRACE_BLOCK];
                        if (late)
                            TailQ::plane(tail.resume, C, 2)[ridx & 0x7fffffffu] = 0xffffffffu; // its old record: moved on, nothing to shade there
                        else
                            rq.ids()[ridx] = idw | SKH_PARKED_BIT; // k_shade leaves this queue entry alone
                        hasRay = false; // (no result of its own: `pending` stays false)
                        parked = true;
                    }
                }
                if (__any(parked))
                    continue; // (the top of the loop writes the results of the lanes that had finished before, if all 64 lanes are idle now)
            }
        }
        if (!__any(hasRay))
        {
            if (exhausted)
                break;
            continue;
        }
#if SKH_CURVE_COOP
        if constexpr (CURVES)
        {
            // ---- the iterative curve intersector, wave-cooperative ----
            // A lane whose leaf produced candidates (segments that passed the cylinder test) parks; each candidate needs two independent
            // Newton runs (one from either end of the segment, up to 40 steps of ~70 instructions).  Run by their owners, the wave waited
            // for 48 parked lanes before it started them (a block costs the same for 3 lanes as for 64) and then ran two to four runs per
            // lane back to back: on average half of the wave sat parked (hair stand-in: 13 of 64 lanes per VALU instruction).  Here the
            // runs of all parked lanes are dealt out one per lane -- to EVERY lane, idle and descending ones included --, so ~25 parked
            // lanes already fill the wave and a block lasts one run.  A run sees the owner's ray (pulled with ds_bpermute) and returns
            // (t, u) or nothing; the owner applies the interval's upper end, takes the nearer root of a candidate (the first run's on a
            // tie) and merges candidates in slot order -- the same decisions in the same order as intersect_curve_segment, same bits.
            __shared__ uint16_t s_runs[256]; // run -> owner lane | slot << 6 | end << 9
            // (a block takes at most two candidates per lane -- a curve leaf holds at most two sub-segments; anything beyond waits for the next block)
            const uint32_t take = hasRay ? ((pend & (0u - pend)) | ((pend & (pend - 1u)) & (0u - (pend & (pend - 1u))))) : 0u;
            const uint32_t myCand = (uint32_t)__popc(take);
            uint32_t incl = 2u * myCand; // inclusive prefix sum of the run counts
#pragma unroll
            for (int off = 1; off < 64; off <<= 1)
            {
                const uint32_t v = __shfl_up(incl, off);
                incl += lane >= (uint32_t)off ? v : 0u;
            }
            const uint32_t nRuns = (uint32_t)__shfl(incl, 63);
            const uint32_t nWalking = (uint32_t)__popcll(__ballot(hasRay && pend == 0u));
            if (nRuns != 0u && (nRuns >= curveMin || nWalking == 0u))
            {
                const uint32_t P = incl - 2u * myCand;
                {
                    uint32_t bits = take, j = P; // (<= 64 lanes x 2 candidates x 2 ends = 256 runs)
                    while (bits != 0u)
                    {
                        const uint32_t k = (uint32_t)__ffs((int)bits) - 1u;
                        bits &= bits - 1u;
                        s_runs[j] = (uint16_t)(lane | (k << 6));
                        s_runs[j + 1u] = (uint16_t)(lane | (k << 6) | (1u << 9));
                        j += 2u;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t myFirst = ((uint32_t)~cur) >> 3; // (a parked lane keeps its leaf in `cur`)
                uint32_t bitsLeft = take, doneCand = 0;
                for (uint32_t base = 0; base < nRuns; base += 64u)
                {
                    const uint32_t r = base + lane;
                    const bool work = r < nRuns;
                    const uint32_t desc = work ? (uint32_t)s_runs[r & 255u] : 0u;
                    const int owner = (int)(desc & 63u);
                    const uint32_t slot = (desc >> 6) & 7u, ep = desc >> 9;
                    // the owner's (object-space) ray and leaf
                    const v3 oo = mk3(__shfl(o.x, owner), __shfl(o.y, owner), __shfl(o.z, owner));
                    const v3 od = mk3(__shfl(d.x, owner), __shfl(d.y, owner), __shfl(d.z, owner));
                    const float otmin = __shfl(tmin, owner);
                    const uint32_t ofirst = (uint32_t)__shfl((int)myFirst, owner);
                    float resT = 0.0f, resU = -1.0f;
                    if (work)
                    {
                        const float4* cp = sc.segs + 4 * (size_t)(ofirst + slot);
                        const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
                        const float dlen = sqrtf(dot(od, od));
                        const float inv_dlen = 1.0f / dlen;
                        const v3 dn = od * inv_dlen;
                        v3 bx, by;
                        onb_from_z(dn, bx, by);
                        CubicPoly poly;
                        {
                            v4 qc[4];
                            const v3 p0 = mk3(c0.x, c0.y, c0.z) - oo, p1 = mk3(c1.x, c1.y, c1.z) - oo, p2 = mk3(c2.x, c2.y, c2.z) - oo, p3 = mk3(c3.x, c3.y, c3.z) - oo;
                            qc[0] = mk4(dot(p0, bx), dot(p0, by), dot(p0, dn), c0.w);
                            qc[1] = mk4(dot(p1, bx), dot(p1, by), dot(p1, dn), c1.w);
                            qc[2] = mk4(dot(p2, bx), dot(p2, by), dot(p2, dn), c2.w);
                            qc[3] = mk4(dot(p3, bx), dot(p3, by), dot(p3, dn), c3.w);
                            cubic_from_bspline(poly, qc);
                        }
                        const v4 e0 = cubic_position(poly, 0.0f);
                        const v4 e1 = cubic_position(poly, 1.0f);
                        const float tstart = (e1.z - e0.z) > 0.0f ? 0.0f : 1.0f;
                        float tpar = ep == 0u ? tstart : 1.0f - tstart;
                        float told = 0.0f, dt1 = 0.0f, dt2 = 0.0f;
                        for (int it = 0; it < 40; ++it)
                        {
                            // one step of the ray / tangent-cone iteration (intersect_curve_segment, loop body)
                            const v4 c4 = cubic_position(poly, tpar);
                            const v4 d4 = ((3.0f * poly.p[0] * tpar) + 2.0f * poly.p[1]) * tpar + poly.p[2];
                            const v3 cc0 = mk3(c4), cd = mk3(d4);
                            const float rr = c4.w, dr = d4.w;
                            const float r2 = rr * rr;
                            const float drr = rr * dr;
                            float ddd = cd.x * cd.x + cd.y * cd.y;
                            const float dp = cc0.x * cc0.x + cc0.y * cc0.y;
                            const float cdd = cc0.x * cd.x + cc0.y * cd.y;
                            const float cxd = cc0.x * cd.y - cc0.y * cd.x;
                            const float cc = ddd;
                            const float bb = cd.z * (drr - cdd);
                            const float cdz2 = cd.z * cd.z;
                            ddd += cdz2;
                            const float aa = ((2.0f * drr * cdd + cxd * cxd) - ddd * r2) + dp * cdz2;
                            const float det = bb * bb - aa * cc;
                            const float ss = (bb - (det > 0.0f ? sqrtf(det) : 0.0f)) / cc;
                            float dt = (ss * cd.z - cdd) / ddd;
                            const bool phantom = !(det > 0.0f);
                            if (!phantom && fabsf(dt) < 5e-5f)
                            {
                                const float sw = (ss + cc0.z) * inv_dlen;
                                if (sw > otmin && tpar >= 0.0f && tpar <= 1.0f) // (the upper end of the interval is the owner's to apply)
                                {
                                    resT = sw;
                                    resU = tpar;
                                }
                                break;
                            }
                            if (phantom && fabsf(dt) < 5e-5f)
                                break; // converged onto a point the ray does not touch: a miss (intersect_curve_segment's rule)
                            dt = fminf(dt, 0.5f);
                            dt = fmaxf(dt, -0.5f);
                            dt1 = dt2;
                            dt2 = dt;
                            if (dt1 * dt2 < 0.0f)
                            {
                                float tnext;
                                if ((it & 3) == 0)
                                    tnext = 0.5f * (told + tpar);
                                else
                                    tnext = (dt2 * told - dt1 * tpar) / (dt2 - dt1);
                                told = tpar;
                                tpar = tnext;
                            }
                            else
                            {
                                told = tpar;
                                tpar += dt;
                            }
#ifdef SKH_ITER_STATS
                            if (COUNT)
                                tc.insts++; // (one-off measurement: Newton steps, reported as "instances")
#endif
                            if (!(tpar >= 0.0f && tpar <= 1.0f))
                                break;
                        }
                    }
                    // owners collect the runs of this round: candidate c of a lane = runs P + 2c (its first end) and P + 2c + 1
                    const uint32_t maxCand = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max(myCand));
                    for (uint32_t c = 0; c < maxCand; ++c)
                    {
                        const uint32_t rr0 = P + 2u * c;
                        const bool mine = c < myCand && c == doneCand && rr0 >= base && rr0 < base + 64u;
                        const int src = (int)((rr0 - base) & 63u);
                        const float t0 = __shfl(resT, src), u0 = __shfl(resU, src);
                        const float t1 = __shfl(resT, (src + 1) & 63), u1 = __shfl(resU, (src + 1) & 63);
                        if (mine)
                        {
                            const uint32_t k = (uint32_t)__ffs((int)bitsLeft) - 1u;
                            bitsLeft &= bitsLeft - 1u;
                            ++doneCand;
                            if (COUNT)
                                tc.segs++;
                            // intersect_curve_segment's acceptance: a root counts if it is within (tmin, tmax], tmax = best.t now; the nearer
                            // root wins, the first run's on a tie
                            const bool f0 = u0 >= 0.0f && t0 <= best.t, f1 = u1 >= 0.0f && t1 <= best.t;
                            const bool first = f0 && !(f1 && t1 < t0);
                            const float t = first ? t0 : t1, u = first ? u0 : u1;
                            if ((f0 || f1) && (best.found || t < best.t)) // (open at tmax: best.t is the ray's tmax until a hit is found)
                            {
                                const uint32_t spw = sc.segPrim[myFirst + k];
                                const uint32_t prim = spw & 0x0fffffffu;
                                // a sub-range leaf keeps the hit only if u is its own (the leaf that owns u reports the same bits)
                                if (min((uint32_t)(u * (float)sc.curveSplit), sc.curveSplit - 1u) == (spw >> 28) &&
                                    (!best.found || t < best.t || curInst < best.inst || (curInst == best.inst && prim < best.prim)))
                                {
                                    best.t = t;
                                    best.inst = curInst;
                                    best.prim = prim;
                                    best.u = u;
                                    best.v = 0.0f;
                                    best.found = true;
                                }
                            }
                        }
                    }
                }
                if (myCand != 0u)
                {
                    pend &= ~take;
                    if (pend == 0u)
                        cur = SKH_REF_INVALID; // leaf done: the lane pops its next entry below
                }
                __builtin_amdgcn_wave_barrier(); // (s_runs is rewritten by the next block)
            }
        }
#endif
        bool terminated = false;
        SKH_LP(uint32_t itN = 0, itT = 0; { const unsigned long long t = __builtin_readcyclecounter(); cy[0] += t - cyA; cyA = t; })
        // (lanes parked in front of the curve intersector do not count: they are not waiting for the node loop to end)
        const uint32_t breakBelow = ((uint32_t)__popcll(__ballot(hasRay && !(CURVES && pend != 0u))) * nodeBreak) >> 6;
        if (hasRay || TRICOOP) // (TRICOOP: every lane comes along to the triangle pass; the node loop and the pop stay with the lanes that have a ray)
        {
            // ---- descend through internal nodes ----
            // (CULL: a lane whose popped entry was culled -- cur INVALID, stack not empty -- stays in the loop, masked for the node block,
            // and pops its next entry at the bottom of the iteration: no inner loop, the chain of culled pops hides behind the other lanes' nodes)
            while ((!TRICOOP || hasRay) && ((cur >= 0 && cur != SKH_REF_INVALID) || (CULL && cur == SKH_REF_INVALID && sp > 0)))
            {
                if (!CULL || cur != SKH_REF_INVALID)
                {
                SKH_LP(itN++; rayNodes++;)
                if constexpr (W8)
                {
                    // one 96-byte fetch = eight quantised child boxes; slot order is traversal order (Node8, skh_bvh.h): no sorting network
                    const float4* np = reinterpret_cast<const float4*>(nodes) + 6 * (size_t)cur;
                    const float4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3], w4 = np[4], w5 = np[5];
                    if (COUNT)
                        tc.nodes++;
                    SKH_LP(if (!inBlas) tc.segs++;)
                    const uint32_t ex = __float_as_uint(w0.w);
                    const float ax = __uint_as_float((ex & 0xffu) << 23) * inv.x, bx = (w0.x - o.x) * inv.x;
                    const float ay = __uint_as_float((ex & 0xff00u) << 15) * inv.y, by = (w0.y - o.y) * inv.y;
                    const float az = __uint_as_float((ex & 0xff0000u) << 7) * inv.z, bz = (w0.z - o.z) * inv.z;
                    const bool px = inv.x >= 0.0f, py = inv.y >= 0.0f, pz = inv.z >= 0.0f;
                    uint32_t nxw[2], fxw[2], nyw[2], fyw[2], nzw[2], fzw[2];
                    nxw[0] = __float_as_uint(px ? w1.x : w1.z), nxw[1] = __float_as_uint(px ? w1.y : w1.w);
                    fxw[0] = __float_as_uint(px ? w1.z : w1.x), fxw[1] = __float_as_uint(px ? w1.w : w1.y);
                    nyw[0] = __float_as_uint(py ? w2.x : w2.z), nyw[1] = __float_as_uint(py ? w2.y : w2.w);
                    fyw[0] = __float_as_uint(py ? w2.z : w2.x), fyw[1] = __float_as_uint(py ? w2.w : w2.y);
                    nzw[0] = __float_as_uint(pz ? w3.x : w3.z), nzw[1] = __float_as_uint(pz ? w3.y : w3.w);
                    fzw[0] = __float_as_uint(pz ? w3.z : w3.x), fzw[1] = __float_as_uint(pz ? w3.w : w3.y);
                    int r[8];
                    r[0] = __float_as_int(w4.x), r[1] = __float_as_int(w4.y), r[2] = __float_as_int(w4.z), r[3] = __float_as_int(w4.w);
                    r[4] = __float_as_int(w5.x), r[5] = __float_as_int(w5.y), r[6] = __float_as_int(w5.z), r[7] = __float_as_int(w5.w);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                    {
                        const int h = k >> 2, sft = 8 * (k & 3);
                        const float nx = fmaf((float)((nxw[h] >> sft) & 0xffu), ax, bx), fx = fmaf((float)((fxw[h] >> sft) & 0xffu), ax, bx);
                        const float ny = fmaf((float)((nyw[h] >> sft) & 0xffu), ay, by), fy = fmaf((float)((fyw[h] >> sft) & 0xffu), ay, by);
                        const float nz = fmaf((float)((nzw[h] >> sft) & 0xffu), az, bz), fz = fmaf((float)((fzw[h] >> sft) & 0xffu), az, bz);
                        const float tnear = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
                        const float tfar = fminf(fminf(fx, fy), fminf(fz, best.t));
                        r[k] = tnear <= tfar * SKH_SLAB_SLACK ? r[k] : SKH_REF_INVALID;
                    }
                    if (!ANY_HIT)
                    {
                        // visit order k <-> slot k ^ oct, oct = signs of the direction: three conditional butterfly stages
#define SKH_BFLY(a, b, keep)                  \
    {                                         \
        const int ta = keep ? r[a] : r[b];    \
        const int tb = keep ? r[b] : r[a];    \
        r[a] = ta, r[b] = tb;                 \
    }
                        SKH_BFLY(0, 1, px) SKH_BFLY(2, 3, px) SKH_BFLY(4, 5, px) SKH_BFLY(6, 7, px)
                        SKH_BFLY(0, 2, py) SKH_BFLY(1, 3, py) SKH_BFLY(4, 6, py) SKH_BFLY(5, 7, py)
                        SKH_BFLY(0, 4, pz) SKH_BFLY(1, 5, pz) SKH_BFLY(2, 6, pz) SKH_BFLY(3, 7, pz)
#undef SKH_BFLY
                    }
                    // the hit children go on the stack last-to-visit first; the first-to-visit one (the last written) is taken back as `cur`
                    int top = SKH_REF_INVALID;
                    if (sp + 8 <= SKH_STACK_LDS)
                    {
                        int* p = lds + sp * SKH_TRACE_BLOCK;
#pragma unroll
                        for (int k = 7; k >= 0; --k)
                        {
                            const bool v = r[k] != SKH_REF_INVALID;
                            *p = r[k]; // (unconditional: what lands above the top is never read)
                            top = v ? r[k] : top;
                            p += v ? SKH_TRACE_BLOCK : 0;
                            sp += v ? 1 : 0;
                        }
                    }
                    else
                    {
#pragma unroll
                        for (int k = 7; k >= 0; --k)
                            if (r[k] != SKH_REF_INVALID)
                            {
                                SKH_PUSH(r[k]);
                                top = r[k];
                            }
                    }
                    if (top != SKH_REF_INVALID)
                        --sp;
                    cur = top;
                }
                else
                {
                // one 64-byte fetch = four quantised child boxes
                const float4* np = reinterpret_cast<const float4*>((WORLD ? sc.triNodes : nodes) + cur);
                const float4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
                if (PF2)
                {
                    // issued AFTER this node's four loads: vector loads return in order, so the wait for the node (vmcnt(1)) leaves the touch in
                    // flight; its value is "used" one iteration later, when it has long arrived behind that iteration's node
                    asm volatile("" ::"v"(pfv));
                    // branch-free (a conditional touch would make the wait for the node cover it too): no candidate = this node's own line again
                    const bool isl = pf < 0;
                    const uint32_t idx = isl ? (((uint32_t)~pf) >> 3) * 3u : (uint32_t)pf * 4u; // in 16-byte units
                    const float4* base = isl ? sc.tris : reinterpret_cast<const float4*>(sc.triNodes);
                    const float4* ta = pf != SKH_REF_INVALID ? base + idx : np;
                    pfv = *reinterpret_cast<const int*>(ta); // (kept alive by the asm above, one iteration later)
                }
                if (COUNT)
                    tc.nodes++;
                SKH_LP(if (!inBlas) tc.segs++;) // (profile build: TLAS share of the node visits, reported as "segs")
                // per axis: plane t = q * (cell * inv) + (o_node - o_ray) * inv (cell sizes come as floats); near/far bytes picked by the sign of inv
                const float ax = w1.w * inv.x, bx = (w0.x - o.x) * inv.x;
                const float ay = w2.w * inv.y, by = (w0.y - o.y) * inv.y;
                const float az = w0.w * inv.z, bz = (w0.z - o.z) * inv.z;
                const bool px = inv.x >= 0.0f, py = inv.y >= 0.0f, pz = inv.z >= 0.0f;
                const uint32_t nxw = __float_as_uint(px ? w1.x : w2.x), fxw = __float_as_uint(px ? w2.x : w1.x);
                const uint32_t nyw = __float_as_uint(py ? w1.y : w2.y), fyw = __float_as_uint(py ? w2.y : w1.y);
                const uint32_t nzw = __float_as_uint(pz ? w1.z : w2.z), fzw = __float_as_uint(pz ? w2.z : w1.z);
                float tn[4];
                int rf[4];
                rf[0] = __float_as_int(w3.x), rf[1] = __float_as_int(w3.y), rf[2] = __float_as_int(w3.z), rf[3] = __float_as_int(w3.w);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                {
#if SKH_PK_NODE
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const f2 qx = { (float)((nxw >> (8 * k)) & 0xffu), (float)((fxw >> (8 * k)) & 0xffu) };
                    const f2 qy = { (float)((nyw >> (8 * k)) & 0xffu), (float)((fyw >> (8 * k)) & 0xffu) };
                    const f2 qz = { (float)((nzw >> (8 * k)) & 0xffu), (float)((fzw >> (8 * k)) & 0xffu) };
                    const f2 tx = __builtin_elementwise_fma(qx, (f2){ ax, ax }, (f2){ bx, bx });
                    const f2 ty = __builtin_elementwise_fma(qy, (f2){ ay, ay }, (f2){ by, by });
                    const f2 tz = __builtin_elementwise_fma(qz, (f2){ az, az }, (f2){ bz, bz });
                    const float nx = tx.x, fx = tx.y, ny = ty.x, fy = ty.y, nz = tz.x, fz = tz.y;
#else
                    const float nx = fmaf((float)((nxw >> (8 * k)) & 0xffu), ax, bx), fx = fmaf((float)((fxw >> (8 * k)) & 0xffu), ax, bx);
                    const float ny = fmaf((float)((nyw >> (8 * k)) & 0xffu), ay, by), fy = fmaf((float)((fyw >> (8 * k)) & 0xffu), ay, by);
                    const float nz = fmaf((float)((nzw >> (8 * k)) & 0xffu), az, bz), fz = fmaf((float)((fzw >> (8 * k)) & 0xffu), az, bz);
#endif
                    const float tnear = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
                    const float tfar = fminf(fminf(fx, fy), fminf(fz, best.t));
                    // (an empty slot is stored as the inverted box 255 > 0 on every axis and fails this test by itself; should rounding
                    // ever let one through, its SKH_REF_INVALID is pushed and skipped when popped)
                    const bool hit = tnear <= tfar * SKH_SLAB_SLACK;
                    tn[k] = hit ? tnear : INFINITY;
                }
                // sort the four candidates by entry distance (5-comparator network), nearest first
#define SKH_CSWAP(a, b)                      \
    {                                        \
        const bool sw = tn[b] < tn[a];       \
        const float ta = sw ? tn[b] : tn[a]; \
        const float tb = sw ? tn[a] : tn[b]; \
        const int ra = sw ? rf[b] : rf[a];   \
        const int rb = sw ? rf[a] : rf[b];   \
        tn[a] = ta, tn[b] = tb;              \
        rf[a] = ra, rf[b] = rb;              \
    }
                if (!ANY_HIT || SKH_SORT_ANYHIT)
                {
                    SKH_CSWAP(0, 1)
                    SKH_CSWAP(2, 3)
                    SKH_CSWAP(0, 2)
                    SKH_CSWAP(1, 3)
                    SKH_CSWAP(1, 2)
                    if (sp + 3 <= NLDS)
                    {
                        // the c hit children among rf[1..3] go to slots sp .. sp+c-1 (farthest first); the writes are
                        // unconditional (what lands above the new top is never read): no branch per push
                        const int c = (tn[1] < INFINITY ? 1 : 0) + (tn[2] < INFINITY ? 1 : 0) + (tn[3] < INFINITY ? 1 : 0);
                        if (CULL)
                        {
                            int2* p = lds2 + sp * SKH_TRACE_BLOCK;
                            const int2 e1 = make_int2(rf[1], __float_as_int(tn[1])), e2 = make_int2(rf[2], __float_as_int(tn[2])), e3 = make_int2(rf[3], __float_as_int(tn[3]));
                            p[0] = c == 3 ? e3 : (c == 2 ? e2 : e1);
                            p[SKH_TRACE_BLOCK] = c == 3 ? e2 : e1;
                            p[2 * SKH_TRACE_BLOCK] = e1;
                        }
                        else
                        {
                            int* p = lds + sp * SKH_TRACE_BLOCK;
                            p[0] = c == 3 ? rf[3] : (c == 2 ? rf[2] : rf[1]);
                            p[SKH_TRACE_BLOCK] = c == 3 ? rf[2] : rf[1];
                            p[2 * SKH_TRACE_BLOCK] = rf[1];
                        }
                        sp += c;
                    }
                    else
                    {
                        if (tn[3] < INFINITY)
                            SKH_PUSH_T(rf[3], __float_as_int(tn[3]));
                        if (tn[2] < INFINITY)
                            SKH_PUSH_T(rf[2], __float_as_int(tn[2]));
                        if (tn[1] < INFINITY)
                            SKH_PUSH_T(rf[1], __float_as_int(tn[1]));
                    }
                    cur = tn[0] < INFINITY ? rf[0] : SKH_REF_INVALID;
                    if (PF2)
                        pf = tn[1] < INFINITY ? rf[1] : SKH_REF_INVALID;
                }
                else
                {
                    // occlusion query: any order finds an occluder; skip the ordering network
                    cur = SKH_REF_INVALID;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (tn[k] < INFINITY)
                        {
                            if (cur != SKH_REF_INVALID)
                                SKH_PUSH(cur);
                            cur = rf[k];
                        }
                }
#undef SKH_CSWAP
                }
                }
                // a lane whose node had no hit child takes its next stack entry right here instead of idling until
                // the whole wave leaves the node loop
                if (cur == SKH_REF_INVALID && sp > 0)
                    SKH_POP(cur); // (CULL: may come back culled = INVALID again; the lane then sits out one iteration and pops the next entry)
                if (POSTPONE && cur < 0 && leaf2 == SKH_REF_INVALID)
                {
                    // the first leaf is put aside and the lane goes on with its next stack entry; closest hit = min over all primitives with a
                    // key tie-break, so the order of the tests cannot change a result -- only which boxes the shrinking best.t still culls
                    leaf2 = cur;
                    cur = SKH_REF_INVALID;
                    if (sp > 0)
                        SKH_POP(cur);
                }
                // few lanes still descending while the rest wait at their leaves: let the leaves go first
                if ((uint32_t)__popcll(__ballot(cur >= 0 && cur != SKH_REF_INVALID)) < breakBelow)
                    break;
            }
            if (!WORLD && cur == SKH_REF_SENTINEL)
            {
                o = ow;
                d = dw;
                inv = ANY_HIT ? invw : rcp3(dw);
                nodes = sc.tlasNodes;
                inBlas = false;
                cur = SKH_REF_INVALID;
            }
#ifdef SKH_EXCHANGE_PROBE
            // (measurement only, docs/LOG.md "re-binning": the LDS traffic of handing a ray to another lane at every phase change --
            // SKH_EXCHANGE_PROBE dwords of state out and back through the dead part of the lane's own stack column, no result changes)
            {
                volatile int* xs = lds;
                float* st[16] = { &o.x, &o.y, &o.z, &inv.x, &inv.y, &inv.z, &tmin, &best.t, &best.u, &best.v, &sh.Sx, &sh.Sy, &sh.Sz, &d.x, &d.y, &d.z };
                const int top = sp < SKH_STACK_LDS - SKH_EXCHANGE_PROBE ? sp : 0; // (a full stack: the probe borrows the bottom, and restores it)
                int saved[SKH_EXCHANGE_PROBE];
#pragma unroll
                for (int k = 0; k < SKH_EXCHANGE_PROBE; ++k)
                {
                    saved[k] = xs[(top + k) * SKH_TRACE_BLOCK];
                    xs[(top + k) * SKH_TRACE_BLOCK] = __float_as_int(*st[k]);
                }
#pragma unroll
                for (int k = 0; k < SKH_EXCHANGE_PROBE; ++k)
                {
                    *st[k] = __int_as_float(xs[(top + k) * SKH_TRACE_BLOCK]);
                    xs[(top + k) * SKH_TRACE_BLOCK] = saved[k];
                }
            }
#endif
            // ---- leaf ----
            SKH_LP({ const unsigned long long t = __builtin_readcyclecounter(); cy[1] += t - cyA; cyA = t; })
            bool entered = false;
            // Two kinds of leaf work (instance entry, primitive tests) are two branches of the same wave.  When one of them has
            // only a few takers it is postponed: those lanes keep their leaf and meet the next pass's takers (leafMin = 0/1: off)
            bool isLeaf = (!TRICOOP || hasRay) && cur < 0 && cur != SKH_REF_SENTINEL;
            if (CURVES)
            {
                // The iterative curve intersector costs ~1000 instructions; run for the one or two lanes that happen to need it, it
                // owns the wave (measured on the hair stand-in: 89 % of the kernel time at ~3 active lanes).  Lanes whose segment
                // passed the cheap cylinder test PARK in front of it (`pend`) and the block runs once `curveMin` lanes wait, or
                // when no other lane of the wave can make progress.
#if SKH_CURVE_COOP
                if (pend != 0u)
                {
                    isLeaf = false;
                    entered = true; // parked: waits for the cooperative block at the top of the loop (no pop)
                }
#else
                const uint32_t nParked = (uint32_t)__popcll(__ballot(pend != 0u)), nActive = (uint32_t)__popcll(__ballot(pend == 0u));
                if (pend != 0u)
                {
                    isLeaf = false;
                    if (nParked >= curveMin || nActive == 0u)
                    {
                        const uint32_t first = ((uint32_t)~cur) >> 3;
                        for (uint32_t k = 0; k < 8u; ++k)
                        {
                            if (!((pend >> k) & 1u))
                                continue;
                            const float4* cp = sc.segs + 4 * (size_t)(first + k);
                            const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
                            if (COUNT)
                                tc.segs++;
                            v4 q[4];
                            q[0] = mk4(c0.x, c0.y, c0.z, c0.w);
                            q[1] = mk4(c1.x, c1.y, c1.z, c1.w);
                            q[2] = mk4(c2.x, c2.y, c2.z, c2.w);
                            q[3] = mk4(c3.x, c3.y, c3.z, c3.w);
                            float t, u;
                            if (intersect_curve_segment(o, d, tmin, best.t, q, t, u) && (best.found || t < best.t)) // (open at tmax: best.t is the ray's tmax until a hit is found)
                            {
                                const uint32_t sp = sc.segPrim[first + k];
                                const uint32_t prim = sp & 0x0fffffffu;
                                // a sub-range leaf keeps the hit only if u is its own (the leaf that owns u reports the same bits)
                                if (min((uint32_t)(u * (float)sc.curveSplit), sc.curveSplit - 1u) != (sp >> 28))
                                    continue;
                                if (!best.found || t < best.t || curInst < best.inst || (curInst == best.inst && prim < best.prim))
                                {
                                    best.t = t;
                                    best.inst = curInst;
                                    best.prim = prim;
                                    best.u = u;
                                    best.v = 0.0f;
                                    best.found = true;
                                }
                            }
                        }
                        pend = 0u; // leaf done: falls through to the pop below
                    }
                    else
                        entered = true; // keep waiting (no pop)
                }
#endif
            }
            if (!WORLD && leafMin > 1u)
            {
                const uint32_t nI = (uint32_t)__popcll(__ballot(isLeaf && !inBlas)), nT = (uint32_t)__popcll(__ballot(isLeaf && inBlas));
                const bool runI = nI >= nT || nI >= leafMin, runT = nT > nI || nT >= leafMin;
                if (isLeaf && !(inBlas ? runT : runI))
                {
                    isLeaf = false;
                    entered = true; // (keeps `cur`: no pop)
                }
            }
            uint32_t kStart = 0; // (TRICOOP) triangles of this lane's leaf the shared pass has dealt with
            if constexpr (TRICOOP)
            {
                // ---- the shared triangle pass ----
                // Lanes at a leaf test its first triangle; lanes that are NOT at a leaf (descending ones taken out of the node loop, lanes without a
                // ray) test the SECOND triangle of the two-triangle leaves, one each, in the same instructions: a helper parks its own o / shear /
                // tmin / best.t in the free part of its LDS stack column, pulls the owner's with ds_bpermute, and restores.  The owner merges its own
                // result first, then the helper's, by the rule of the sequential loop -- a candidate is accepted if it is nearer than the best hit
                // so far, or equally near with the smaller (instance, primitive) key -- the helper's test only saw a STALE, i.e. larger, tmax, so it
                // reports every candidate the sequential loop could have accepted: same records, bit for bit.
                uint32_t cfirst = 0, ccount = 0;
                if (isLeaf)
                {
                    const uint32_t e = (uint32_t)~cur;
                    cfirst = e >> 3, ccount = (e & 7u) + 1u;
                }
                const bool has2 = ccount >= 2u;
                const int freeFrom = hasRay ? sp : 0;
                const bool canHelp = !isLeaf && (ANY_HIT || freeFrom + 9 <= NLDS);
                const unsigned long long mB = __ballot(has2), mI = __ballot(canHelp);
                const uint32_t nH = min((uint32_t)__popcll(mB), (uint32_t)__popcll(mI));
                bool helped = false, helper = false;
                int partner = (int)lane;
                if (nH != 0u)
                {
                    const uint32_t rankB = rank_below(mB), rankI = rank_below(mI);
                    helped = has2 && rankB < nH;
                    helper = canHelp && rankI < nH;
                    if (helped)
                        s_tab[rankB] = (unsigned char)lane;
                    if (helper)
                        s_tab[64u + rankI] = (unsigned char)lane;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    partner = helped ? (int)s_tab[64u + rankB] : (helper ? (int)s_tab[rankI] : (int)lane);
                }
                // Where the helper keeps the owner's ray: the any-hit build has registers to spare (63 + 9 <= 72: a copy, nothing to park); the
                // closest-hit build sits at its 72-VGPR limit, so there a helper OVERWRITES its own o / shear / tmin / best.t after parking them
                // in the free part of its LDS stack column, and restores them after the test.
                constexpr bool COOP_REGS = ANY_HIT;
                int* park = lds + freeFrom * SKH_TRACE_BLOCK;
                if (!COOP_REGS && helper)
                {
                    park[0] = __float_as_int(o.x), park[SKH_TRACE_BLOCK] = __float_as_int(o.y), park[2 * SKH_TRACE_BLOCK] = __float_as_int(o.z);
                    park[3 * SKH_TRACE_BLOCK] = sh.perm, park[4 * SKH_TRACE_BLOCK] = __float_as_int(sh.Sx), park[5 * SKH_TRACE_BLOCK] = __float_as_int(sh.Sy);
                    park[6 * SKH_TRACE_BLOCK] = __float_as_int(sh.Sz), park[7 * SKH_TRACE_BLOCK] = __float_as_int(tmin), park[8 * SKH_TRACE_BLOCK] = __float_as_int(best.t);
                }
                uint32_t triIdx = cfirst;
                v3 to = o;
                RayShear tsh = sh;
                float ttmin = tmin, ttmax = best.t;
                if (nH != 0u)
                {
                    // (every lane takes part in the exchange: a disabled source lane would read as zero)
                    const float pox = __shfl(o.x, partner), poy = __shfl(o.y, partner), poz = __shfl(o.z, partner);
                    const int pperm = __shfl(sh.perm, partner);
                    const float psx = __shfl(sh.Sx, partner), psy = __shfl(sh.Sy, partner), psz = __shfl(sh.Sz, partner);
                    const float ptmin = __shfl(tmin, partner), pbt = __shfl(best.t, partner);
                    const uint32_t pfirst = (uint32_t)__shfl((int)cfirst, partner);
                    if (helper)
                    {
                        if (COOP_REGS)
                        {
                            to = mk3(pox, poy, poz);
                            tsh.perm = pperm, tsh.Sx = psx, tsh.Sy = psy, tsh.Sz = psz;
                            ttmin = ptmin, ttmax = pbt;
                        }
                        else
                        {
                            o = mk3(pox, poy, poz);
                            sh.perm = pperm, sh.Sx = psx, sh.Sy = psy, sh.Sz = psz;
                            tmin = ptmin;
                            best.t = pbt;
                        }
                        triIdx = pfirst + 1u;
                    }
                }
                bool ih = false;
                float ht = 0.0f, hu = 0.0f, hv = 0.0f;
                uint32_t hprim = 0, hinst = 0;
                if (isLeaf || helper)
                {
                    const float4* tp = sc.tris + 3 * (size_t)triIdx;
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    if (COUNT)
                        tc.prims++;
                    if (COOP_REGS)
                        ih = intersect_triangle(to, tsh, ttmin, ttmax, mk3(a), mk3(b), mk3(c), ht, hu, hv);
                    else
                        ih = intersect_triangle(o, sh, tmin, best.t, mk3(a), mk3(b), mk3(c), ht, hu, hv);
                    hprim = __float_as_uint(a.w), hinst = __float_as_uint(b.w);
                }
                if (!COOP_REGS && helper)
                {
                    o = mk3(__int_as_float(park[0]), __int_as_float(park[SKH_TRACE_BLOCK]), __int_as_float(park[2 * SKH_TRACE_BLOCK]));
                    sh.perm = park[3 * SKH_TRACE_BLOCK], sh.Sx = __int_as_float(park[4 * SKH_TRACE_BLOCK]), sh.Sy = __int_as_float(park[5 * SKH_TRACE_BLOCK]);
                    sh.Sz = __int_as_float(park[6 * SKH_TRACE_BLOCK]), tmin = __int_as_float(park[7 * SKH_TRACE_BLOCK]), best.t = __int_as_float(park[8 * SKH_TRACE_BLOCK]);
                }
#define SKH_MERGE_HIT(H, T, U, V, PRIM, INST)                                                                                         \
    if ((H) && (best.found ? ((T) < best.t || ((T) == best.t && ((INST) < best.inst || ((INST) == best.inst && (PRIM) < best.prim)))) \
                           : (T) < best.t)) /* (open at tmax: best.t is the ray's tmax until a hit is found) */                       \
    {                                                                                                                                 \
        best.t = (T), best.inst = (INST), best.prim = (PRIM), best.u = (U), best.v = (V), best.found = true;                          \
    }
                if (isLeaf)
                    SKH_MERGE_HIT(ih, ht, hu, hv, hprim, hinst)
                kStart = 1u;
                if (nH != 0u)
                {
                    const bool rh = __shfl((int)ih, partner) != 0;
                    const float rt = __shfl(ht, partner), ru = __shfl(hu, partner), rv = __shfl(hv, partner);
                    const uint32_t rprim = (uint32_t)__shfl((int)hprim, partner), rinst = (uint32_t)__shfl((int)hinst, partner);
                    if (helped)
                    {
                        SKH_MERGE_HIT(rh, rt, ru, rv, rprim, rinst)
                        kStart = 2u;
                    }
                }
#undef SKH_MERGE_HIT
            }
            if (isLeaf || (POSTPONE && leaf2 != SKH_REF_INVALID))
            {
                // (POSTPONE: up to two leaves wait here -- the one put aside in the node loop first, then the current one)
                const int leafA = (POSTPONE && leaf2 != SKH_REF_INVALID) ? leaf2 : cur;
                const uint32_t enc = (uint32_t)~leafA;
                const uint32_t first = enc >> 3, count = (enc & 7u) + 1u;
                const bool two = POSTPONE && leaf2 != SKH_REF_INVALID && isLeaf;
                const uint32_t encB = (uint32_t)~cur;
                const uint32_t firstB = encB >> 3, total = count + (two ? (encB & 7u) + 1u : 0u);
                if (POSTPONE)
                    leaf2 = SKH_REF_INVALID;
                if (!WORLD && !inBlas)
                {
                    // TLAS leaves hold exactly one instance
                    // the whole 64-byte record in one round trip (loading the transform only after the mask test made it two)
                    const float4* ip = reinterpret_cast<const float4*>(sc.tinst + first); // `first` = TLAS leaf number
                    const float4 i0 = ip[0], i1 = ip[1], i2 = ip[2], i3 = ip[3];
                    asm volatile("" ::"v"(i0.x), "v"(i1.x), "v"(i2.x)); // (keeps the three loads above the branch: the compiler sinks them into it)
                    if (__float_as_uint(i3.y) & rayMask)
                    {
                        if (COUNT)
                            tc.insts++;
                        SKH_LP(rayInsts++;)
                        const float m[12] = { i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w };
                        o = xform_point_rel(m, ow);
                        d = xform_vector(m, dw);
                        inv = rcp3(d);
                        sh = make_shear(d);
                        curInst = __float_as_uint(i3.w); // the instance this leaf belongs to
                        curType = __float_as_uint(i3.z);
                        nodes = (CURVES && curType == 2) ? sc.segNodes : sc.triNodes;
                        inBlas = true;
                        SKH_PUSH(SKH_REF_SENTINEL);
                        cur = __float_as_int(i3.x);
                        entered = true;
                    }
                }
                else if (CURVES && curType == 2)
                {
                    for (uint32_t k = 0; k < count; ++k)
                    {
                        // cheap conservative rejection: (distance between the ray's line and the segment's bounding cylinder
                        // axis)^2 = ((A - o) . n)^2 / |n|^2, n = d x axis; a long thin diagonal hair fills a tiny part of its box
                        const float4 b0 = sc.segBound[2 * (size_t)(first + k)], b1 = sc.segBound[2 * (size_t)(first + k) + 1];
                        const v3 w = mk3(b0.x - o.x, b0.y - o.y, b0.z - o.z);
                        const v3 nn = cross(d, mk3(b1.x, b1.y, b1.z));
                        const float n2 = dot(nn, nn), wn = dot(w, nn);
                        const float Rm = b0.w + (fabsf(w.x) + fabsf(w.y) + fabsf(w.z)) * 4e-6f; // cancellation in w . n
                        if (n2 > 1e-12f * dot(d, d) && wn * wn > Rm * Rm * n2 * 1.0001f)
                            continue;
                        pend |= 1u << k;
                    }
                    if (pend != 0u)
                        entered = true; // parks in front of the full intersector (see above); `cur` keeps the leaf
                }
                else
                {
                    for (uint32_t k = (TRICOOP ? kStart : 0u); k < total; ++k)
                    {
                        const float4* tp = sc.tris + 3 * (size_t)((POSTPONE && k >= count) ? firstB + (k - count) : first + k);
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        if (COUNT)
                            tc.prims++;
                        SKH_LP(itT++; rayTris++;)
                        float t, u, v;
#ifdef SKH_LANE_PROFILE
                        uint32_t pf = 0;
                        const bool ih = intersect_triangle(o, sh, tmin, best.t, mk3(a), mk3(b), mk3(c), t, u, v, &pf);
                        wv[6] += __any(pf & 1u) ? 1u : 0u; // triangle passes in which some lane took the fp64 edge-function fallback
                        wv[7] += __any(pf & 4u) ? 1u : 0u; // ... in which some lane got as far as the division
                        wv[8] += __any(pf & 2u) ? 1u : 0u; // ... passed the sign test
                        if (ih && (best.found || t < best.t))
#else
                        if (intersect_triangle(o, sh, tmin, best.t, mk3(a), mk3(b), mk3(c), t, u, v) && (best.found || t < best.t))
#endif
                        {
                            const uint32_t prim = __float_as_uint(a.w);
                            const uint32_t hinst = (WORLD || curInst == 0xffffffffu) ? __float_as_uint(b.w) : curInst; // (baked group: the record names its instance)
                            if (!best.found || t < best.t || hinst < best.inst || (hinst == best.inst && prim < best.prim))
                            {
                                best.t = t;
                                best.inst = hinst;
                                best.prim = prim;
                                best.u = u;
                                best.v = v;
                                best.found = true;
                            }
                        }
                    }
                }
            }
            SKH_LP(if (entered) itT |= 0x10000u; { const unsigned long long t = __builtin_readcyclecounter(); cy[2] += t - cyA; cyA = t; })
            // ---- pop ----
            if (ANY_HIT && (!TRICOOP || hasRay) && best.found)
                terminated = true;
            else if ((!TRICOOP || hasRay) && !entered && !(cur >= 0 && cur != SKH_REF_INVALID)) // (a lane taken out of the node loop early keeps its node)
            {
                for (;;)
                {
                    if (sp == 0)
                    {
                        terminated = true;
                        break;
                    }
                    SKH_POP(cur);
                    if (!WORLD && cur == SKH_REF_SENTINEL)
                    {
                        o = ow;
                        d = dw;
                        inv = ANY_HIT ? invw : rcp3(dw);
                        nodes = sc.tlasNodes;
                        inBlas = false;
                        continue;
                    }
                    break; // (CULL: a culled entry comes back as INVALID; the node loop of the next pass pops on, an empty stack ends the ray below)
                }
            }
        }
#ifdef SKH_LANE_PROFILE
        if (hasRay)
        {
            const unsigned long long t = __builtin_readcyclecounter();
            cy[3] += t - cyA;
        }
        cyA = __builtin_readcyclecounter();
        wv[0] += wave_max(itN);
        wv[1] += wave_max(itT & 0xffffu);
        wv[2] += __any((itT >> 16) != 0) ? 1u : 0u;
#endif
        if (terminated)
        {
            hasRay = false;
            pending = true; // the result stays in registers until the next refill: one write block per refill, not per termination
#ifdef SKH_LANE_PROFILE
            if (rayNodes > 700u)
            {
                const uint32_t k = atomicAdd(&stats->slowCount, 1u);
                if (k < 16u)
                {
                    float* r = stats->slow[k];
                    r[0] = (float)rayNodes, r[1] = (float)rayTris, r[2] = (float)rayInsts, r[3] = ANY_HIT ? 1.0f : 0.0f;
                    r[4] = ow.x, r[5] = ow.y, r[6] = ow.z, r[7] = dw.x, r[8] = dw.y, r[9] = dw.z, r[10] = tmin, r[11] = (ridx & 0x80000000u) ? 0.0f : rq.plane(7)[ridx];
                }
            }
            rayNodes = rayTris = rayInsts = 0;
#endif
        }
        SKH_LP(cy[4] += __builtin_readcyclecounter() - cyA;)
    }
#undef SKH_PUSH
#undef SKH_POP
#undef SKH_OVF_AT
#undef tail
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
        for (int k = 0; k < 6; ++k)
        {
            // cycle sums are wave-uniform increments taken by the lanes that were active: the busiest lane has (nearly) all of them
            uint32_t hi = wave_max((uint32_t)(cy[k] >> 8));
            if (lane == 0)
                atomicAdd(&stats->cyc[ANY_HIT ? 1 : 0][k], (unsigned long long)hi << 8);
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
