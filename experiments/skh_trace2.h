// strelka_hip -- k_trace2: the two-level traversal of k_trace with TWO rays per lane (triangle scenes).
//
// k_trace sits on the VALU issue rate at ~47 % lane utilisation: a lane whose ray waits at a leaf (or for a refill) idles while
// the wave steps through nodes, and the other way round.  Here every lane owns two rays (slots A and B, 128 per wave).  Each
// pass of the outer loop picks the kind of work most lanes can take part in with EITHER of their rays -- a node step, an
// instance entry or a leaf's triangles --, copies that ray into a working set with selects, runs the block and writes the
// changed fields back.  Same arithmetic, same acceptance rules, same results as k_trace (hit records do not depend on the
// traversal order: DESIGN.md "determinism"); the price is ~110 VGPRs (4 waves per SIMD instead of 6, carrying 8 rays per SIMD
// lane instead of 6) and ~40 select instructions per pass.
#pragma once

namespace skh
{

#define SKH_T2_STACK 16 // LDS stack entries per ray: 128 rays x 16 x 4 B = 8 KB per wave, 16 waves per CU
#define SKH_T2_SLOTS 128

struct RaySlot
{
    v3 ow, dw, o, inv;
    float tmin;
    RayShear sh;
    int cur, sp;
    uint32_t curInst, ridx;
    HitRec best;
    bool has, inBlas, pending;
};

SKH_DI void slot_clear(RaySlot& r)
{
    r.ow = r.dw = r.o = r.inv = mk3(0.0f);
    r.tmin = 0.0f;
    r.sh.perm = 0;
    r.sh.Sx = r.sh.Sy = r.sh.Sz = 0.0f;
    r.cur = SKH_REF_INVALID;
    r.sp = 0;
    r.curInst = r.ridx = 0;
    r.best.t = 0.0f, r.best.inst = r.best.prim = 0xffffffffu, r.best.u = r.best.v = 0.0f, r.best.found = false;
    r.has = r.inBlas = r.pending = false;
}

// kind of work a ray waits for: 0 none, 1 node step, 2 instance entry (TLAS leaf), 3 triangles (BLAS leaf)
SKH_DI int slot_phase(const RaySlot& r)
{
    if (!r.has)
        return 0;
    if (r.cur >= 0)
        return 1; // (a live ray never rests on SKH_REF_INVALID: the pop at the end of every block sees to that)
    return r.inBlas ? 3 : 2;
}

#define SKH_T2_SEL(f) (useA ? A.f : B.f)
#define SKH_T2_PUT(f, v) \
    {                    \
        if (useA)        \
            A.f = (v);   \
        else if (useB)   \
            B.f = (v);   \
    }

template <bool ANY_HIT>
__global__ void __launch_bounds__(SKH_TRACE_BLOCK, 4)
    k_trace2(DevScene sc, RayQ rq, const uint32_t* __restrict__ countPtr, uint32_t* __restrict__ fetch, uint32_t fetchArg, HitQ hq,
             PathS ps, const float* __restrict__ contrib, uint32_t contribStride, int* __restrict__ ovfBase)
{
    __shared__ int s_stack[SKH_T2_STACK * SKH_T2_SLOTS];
    const uint32_t fetchMin = fetchArg & 0xffu, nodeBreak = (fetchArg >> 16) & 0xffu;
    const uint32_t lane = threadIdx.x;
    const uint32_t n = *countPtr;
    if (n == 0)
        return;
    const uint32_t perGroup = (((n + 7u) >> 3) + 63u) & ~63u;
    const uint32_t group = blockIdx.x & 7u;
    uint32_t tries = 0;
    bool exhausted = false;
    const uint32_t ovfStride = gridDim.x * SKH_T2_SLOTS;
    const uint32_t rayMask = ANY_HIT ? 1u : 253u;
    RaySlot A, B;
    slot_clear(A);
    slot_clear(B);

    auto write_result = [&](RaySlot& r) {
        if (!r.pending)
            return;
        r.pending = false;
        const uint32_t i = r.ridx;
        if (ANY_HIT)
        {
            if (hq.base)
                hq.base[i] = r.best.found ? 1.0f : -1.0f;
            else if (!r.best.found)
            {
                const uint32_t pid = rq.ids()[i];
                float* rad = ps.base + (size_t)3 * ps.stride;
                rad[pid] += contrib[i];
                rad[pid + ps.stride] += contrib[i + contribStride];
                rad[pid + 2 * (size_t)ps.stride] += contrib[i + 2 * (size_t)contribStride];
            }
        }
        else
        {
            hq.base[i] = r.best.found ? r.best.t : -1.0f;
            reinterpret_cast<uint32_t*>(hq.base)[i + hq.stride] = r.best.inst;
            reinterpret_cast<uint32_t*>(hq.base)[i + 2 * (size_t)hq.stride] = r.best.prim;
            hq.base[i + 3 * (size_t)hq.stride] = r.best.u;
            hq.base[i + 4 * (size_t)hq.stride] = r.best.v;
        }
    };
    auto load_ray = [&](RaySlot& r, uint32_t idx) {
        r.ridx = idx;
        r.ow = mk3(rq.plane(0)[idx], rq.plane(1)[idx], rq.plane(2)[idx]);
        r.dw = mk3(rq.plane(3)[idx], rq.plane(4)[idx], rq.plane(5)[idx]);
        r.tmin = rq.plane(6)[idx];
        r.o = r.ow;
        r.inv = rcp3(r.dw);
        r.inBlas = false;
        r.sp = 0;
        r.cur = sc.tlasRoot;
        r.best.t = rq.plane(7)[idx];
        r.best.inst = r.best.prim = 0xffffffffu;
        r.best.u = r.best.v = 0.0f;
        r.best.found = false;
        r.has = r.cur != SKH_REF_INVALID; // (empty scene: a miss right away)
        r.pending = !r.has;
    };

    for (;;)
    {
        // ---------------- refill empty slots from the queue ----------------
        const unsigned long long needA = __ballot(!A.has), needB = __ballot(!B.has);
        const uint32_t wantA = (uint32_t)__popcll(needA), want = wantA + (uint32_t)__popcll(needB);
        if (want >= fetchMin || want == (uint32_t)SKH_T2_SLOTS)
        {
            write_result(A);
            write_result(B);
            if (!exhausted)
            {
                uint32_t base = 0, count = 0;
                while (tries < 8u)
                {
                    const uint32_t g = (group + tries) & 7u;
                    uint32_t b = 0;
                    if (lane == 0)
                        b = atomicAdd(&fetch[g * SKH_FETCH_STRIDE], want);
                    b = __shfl(b, 0);
                    const uint32_t lo = g * perGroup;
                    const uint32_t hi = min(n, lo + perGroup);
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
                if (tries >= 8u && count == 0)
                    exhausted = true;
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t rankA = (uint32_t)__popcll(needA & below), rankB = wantA + (uint32_t)__popcll(needB & below);
                if (!A.has && rankA < count)
                    load_ray(A, base + rankA);
                if (!B.has && rankB < count)
                    load_ray(B, base + rankB);
            }
        }
        if (!__any(A.has || B.has))
        {
            if (exhausted)
            {
                write_result(A);
                write_result(B);
                break;
            }
            continue;
        }
        // ---------------- pick the block most lanes can join ----------------
        const int pA = slot_phase(A), pB = slot_phase(B);
        const uint32_t cN = (uint32_t)__popcll(__ballot(pA == 1 || pB == 1)), cI = (uint32_t)__popcll(__ballot(pA == 2 || pB == 2)),
                       cT = (uint32_t)__popcll(__ballot(pA == 3 || pB == 3));
        const int phase = (cN >= cI && cN >= cT) ? 1 : (cT >= cI ? 3 : 2);
        const bool useA = pA == phase, useB = !useA && pB == phase;
        if (useA || useB)
        {
            int* lds = s_stack + lane + (useA ? 0u : 64u);
            int* ovf = ovfBase + (blockIdx.x * SKH_T2_SLOTS + lane + (useA ? 0u : 64u));
#define SKH_PUSH(v)                                                   \
    {                                                                 \
        if (sp < SKH_T2_STACK)                                        \
            lds[sp * SKH_T2_SLOTS] = (v);                             \
        else if (sp < SKH_T2_STACK + SKH_STACK_OVF)                   \
            ovf[(size_t)(sp - SKH_T2_STACK) * ovfStride] = (v);       \
        ++sp;                                                         \
    }
#define SKH_POP(dst)                                                  \
    {                                                                 \
        --sp;                                                         \
        if (sp < SKH_T2_STACK)                                        \
            dst = lds[sp * SKH_T2_SLOTS];                             \
        else if (sp < SKH_T2_STACK + SKH_STACK_OVF)                   \
            dst = ovf[(size_t)(sp - SKH_T2_STACK) * ovfStride];       \
        else                                                          \
            dst = SKH_REF_INVALID;                                    \
    }
            // working copy of the fields every block needs
            int cur = SKH_T2_SEL(cur), sp = SKH_T2_SEL(sp);
            v3 o = SKH_T2_SEL(o), inv = SKH_T2_SEL(inv);
            const float tmin = SKH_T2_SEL(tmin);
            bool inBlas = SKH_T2_SEL(inBlas);
            HitRec best;
            best.t = SKH_T2_SEL(best.t);
            best.found = SKH_T2_SEL(best.found);
            bool terminated = false;
            if (phase == 1)
            {
                const uint32_t breakBelow = (cN * nodeBreak) >> 6;
                while (cur >= 0 && cur != SKH_REF_INVALID)
                {
                    const Node4* nodes = inBlas ? sc.triNodes : sc.tlasNodes;
                    const float4* np = reinterpret_cast<const float4*>(nodes + cur);
                    const float4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
                    const uint32_t exps = __float_as_uint(w0.w);
                    const float ax = __uint_as_float((exps & 0xffu) << 23) * inv.x, bx = (w0.x - o.x) * inv.x;
                    const float ay = __uint_as_float(((exps >> 8) & 0xffu) << 23) * inv.y, by = (w0.y - o.y) * inv.y;
                    const float az = __uint_as_float(((exps >> 16) & 0xffu) << 23) * inv.z, bz = (w0.z - o.z) * inv.z;
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
                        const float nx = fmaf((float)((nxw >> (8 * k)) & 0xffu), ax, bx), fx = fmaf((float)((fxw >> (8 * k)) & 0xffu), ax, bx);
                        const float ny = fmaf((float)((nyw >> (8 * k)) & 0xffu), ay, by), fy = fmaf((float)((fyw >> (8 * k)) & 0xffu), ay, by);
                        const float nz = fmaf((float)((nzw >> (8 * k)) & 0xffu), az, bz), fz = fmaf((float)((fzw >> (8 * k)) & 0xffu), az, bz);
                        const float tnear = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
                        const float tfar = fminf(fminf(fx, fy), fminf(fz, best.t));
                        const bool hit = rf[k] != SKH_REF_INVALID && tnear <= tfar * SKH_SLAB_SLACK;
                        tn[k] = hit ? tnear : INFINITY;
                    }
                    if (!ANY_HIT)
                    {
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
                        SKH_CSWAP(0, 1)
                        SKH_CSWAP(2, 3)
                        SKH_CSWAP(0, 2)
                        SKH_CSWAP(1, 3)
                        SKH_CSWAP(1, 2)
#undef SKH_CSWAP
                        if (sp + 3 <= SKH_T2_STACK)
                        {
                            const int c = (tn[1] < INFINITY ? 1 : 0) + (tn[2] < INFINITY ? 1 : 0) + (tn[3] < INFINITY ? 1 : 0);
                            int* p = lds + sp * SKH_T2_SLOTS;
                            p[0] = c == 3 ? rf[3] : (c == 2 ? rf[2] : rf[1]);
                            p[SKH_T2_SLOTS] = c == 3 ? rf[2] : rf[1];
                            p[2 * SKH_T2_SLOTS] = rf[1];
                            sp += c;
                        }
                        else
                        {
                            if (tn[3] < INFINITY)
                                SKH_PUSH(rf[3]);
                            if (tn[2] < INFINITY)
                                SKH_PUSH(rf[2]);
                            if (tn[1] < INFINITY)
                                SKH_PUSH(rf[1]);
                        }
                        cur = tn[0] < INFINITY ? rf[0] : SKH_REF_INVALID;
                    }
                    else
                    {
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
                    if (cur == SKH_REF_INVALID && sp > 0)
                        SKH_POP(cur);
                    if ((uint32_t)__popcll(__ballot(cur >= 0 && cur != SKH_REF_INVALID)) < breakBelow)
                        break;
                }
            }
            else if (phase == 2)
            {
                // TLAS leaf: exactly one instance
                const uint32_t first = ((uint32_t)~cur) >> 3;
                const float4* ip = reinterpret_cast<const float4*>(sc.tinst + first);
                const float4 i3 = ip[3];
                if (__float_as_uint(i3.y) & rayMask)
                {
                    const float4 i0 = ip[0], i1 = ip[1], i2 = ip[2];
                    const float m[12] = { i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w };
                    const v3 ow = SKH_T2_SEL(ow), dw = SKH_T2_SEL(dw);
                    o = xform_point(m, ow);
                    const v3 d = xform_vector(m, dw);
                    inv = rcp3(d);
                    const RayShear sh = make_shear(d);
                    SKH_T2_PUT(sh.perm, sh.perm)
                    SKH_T2_PUT(sh.Sx, sh.Sx)
                    SKH_T2_PUT(sh.Sy, sh.Sy)
                    SKH_T2_PUT(sh.Sz, sh.Sz)
                    SKH_T2_PUT(curInst, __float_as_uint(i3.w))
                    inBlas = true;
                    SKH_PUSH(SKH_REF_SENTINEL);
                    cur = __float_as_int(i3.x);
                }
                else
                    cur = SKH_REF_INVALID;
            }
            else
            {
                const uint32_t enc = (uint32_t)~cur;
                const uint32_t first = enc >> 3, count = (enc & 7u) + 1u;
                RayShear sh;
                sh.perm = SKH_T2_SEL(sh.perm);
                sh.Sx = SKH_T2_SEL(sh.Sx);
                sh.Sy = SKH_T2_SEL(sh.Sy);
                sh.Sz = SKH_T2_SEL(sh.Sz);
                const uint32_t curInst = SKH_T2_SEL(curInst);
                best.inst = SKH_T2_SEL(best.inst);
                best.prim = SKH_T2_SEL(best.prim);
                best.u = SKH_T2_SEL(best.u);
                best.v = SKH_T2_SEL(best.v);
                for (uint32_t k = 0; k < count; ++k)
                {
                    const float4* tp = sc.tris + 3 * (size_t)(first + k);
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    float t, u, v;
                    // (open at tmax: best.t is the ray's tmax until a hit is found)
                    if (intersect_triangle(o, sh, tmin, best.t, mk3(a), mk3(b), mk3(c), t, u, v) && (best.found || t < best.t))
                    {
                        const uint32_t prim = __float_as_uint(a.w);
                        if (!best.found || t < best.t || curInst < best.inst || (curInst == best.inst && prim < best.prim))
                        {
                            best.t = t;
                            best.inst = curInst;
                            best.prim = prim;
                            best.u = u;
                            best.v = v;
                            best.found = true;
                        }
                    }
                }
                SKH_T2_PUT(best.t, best.t)
                SKH_T2_PUT(best.inst, best.inst)
                SKH_T2_PUT(best.prim, best.prim)
                SKH_T2_PUT(best.u, best.u)
                SKH_T2_PUT(best.v, best.v)
                SKH_T2_PUT(best.found, best.found)
                cur = SKH_REF_INVALID;
            }
            // ---- next reference: pop until a node or a leaf, leaving instances on the way ----
            if (ANY_HIT && best.found)
                terminated = true;
            else
                while (cur == SKH_REF_INVALID || cur == SKH_REF_SENTINEL)
                {
                    if (cur == SKH_REF_SENTINEL)
                    {
                        o = SKH_T2_SEL(ow);
                        inv = rcp3(SKH_T2_SEL(dw));
                        inBlas = false;
                    }
                    if (sp == 0)
                    {
                        terminated = true;
                        break;
                    }
                    SKH_POP(cur);
                }
#undef SKH_PUSH
#undef SKH_POP
            // ---- write the changed fields back ----
            SKH_T2_PUT(cur, cur)
            SKH_T2_PUT(sp, sp)
            SKH_T2_PUT(o, o)
            SKH_T2_PUT(inv, inv)
            SKH_T2_PUT(inBlas, inBlas)
            if (terminated)
            {
                SKH_T2_PUT(has, false)
                SKH_T2_PUT(pending, true)
            }
        }
    }
}

#undef SKH_T2_SEL
#undef SKH_T2_PUT

} // namespace skh
