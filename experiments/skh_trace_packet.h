// NOT BUILT -- a measured negative result of round 3, kept for the record (docs/LOG.md "packet traversal for camera rays").
// It was #included from skh_kernels.h and launched for bounce 0 of world-only scenes; bit-identical hit records (92 / 92 GPU tests),
// but closest-hit 86.8 -> 107.4 ms on the kitchen stand-in and 74.0 -> 77.7 ms on its unshared variant.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// k_trace_packet: closest hit for CAMERA rays over the world-only hierarchy, one 64-ray packet per wave.
//
// The first queue of a pass is in slot order (tile-major, Morton inside a tile), so 64 consecutive rays are an 8x8 pixel block of
// one sample: they walk (almost) the same nodes.  The packet walks ONE stack (in LDS, per wave): a node is fetched once for the wave
// (a uniform address: one request, not 64), every lane tests its own ray against the node's four boxes, and a child is visited if
// ANY lane's ray enters it before that lane's nearest hit.  No per-lane stack, no divergence between "descending" and "at a leaf",
// no lane waits for another: all 64 lanes execute every step.  Rays are tested against nodes they would not have reached on their
// own, which is harmless -- boxes only ever have to be conservative, the primitive tests (the same function, the same
// (t, instance, primitive) rule) decide -- so the hit records are bit-identical to k_trace's.
// Visit order: children sorted by the entry distance of the packet's first live lane.
// ------------------------------------------------------------------------------------------------------------
#define SKH_PACKET_STACK 96
__global__ void __launch_bounds__(SKH_TRACE_BLOCK, 8)
    k_trace_packet(DevScene sc, RayQ rq, const uint32_t* __restrict__ countPtr, uint32_t* __restrict__ fetch /*8 cursors, zeroed*/, HitQ hq)
{
    __shared__ int s_pstack[SKH_PACKET_STACK];
    const uint32_t lane = threadIdx.x;
    uint32_t tries = 0;
    const uint32_t group = blockIdx.x & 7u;
    for (;;)
    {
        // ---- next packet: 64 consecutive queue positions of one shard ----
        uint32_t base = 0, count = 0;
        while (tries < 8u)
        {
            const uint32_t g = (group + tries) & 7u;
            uint32_t b = 0;
            if (lane == 0u)
                b = atomicAdd(&fetch[g * SKH_FETCH_STRIDE], 64u);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            const uint32_t n = countPtr[g * SKH_COUNT_STRIDE];
            if (b < n)
            {
                base = g * rq.region + b;
                count = min(64u, n - b);
                break;
            }
            ++tries;
        }
        if (count == 0u)
            return;
        const bool live = lane < count;
        const uint32_t ridx = base + (live ? lane : 0u);
        const v3 o = mk3(rq.plane(0)[ridx], rq.plane(1)[ridx], rq.plane(2)[ridx]);
        const v3 d = mk3(rq.plane(3)[ridx], rq.plane(4)[ridx], rq.plane(5)[ridx]);
        const float tmin = rq.plane(6)[ridx];
        const v3 inv = rcp3(d);
        const RayShear sh = make_shear(d);
        HitRec best;
        best.t = rq.plane(7)[ridx];
        best.inst = best.prim = 0xffffffffu;
        best.u = best.v = 0.0f;
        best.found = false;
        const int firstLane = 0; // (lane 0 is live in every packet)
        int sp = 0;
        int cur = sc.worldRoot;
        if (sc.lightRoot != SKH_REF_INVALID)
        {
            if (cur != SKH_REF_INVALID)
            {
                if (lane == 0u)
                    s_pstack[0] = sc.lightRoot;
                sp = 1;
            }
            else
                cur = sc.lightRoot;
        }
        while (cur != SKH_REF_INVALID)
        {
            if (cur >= 0)
            {
                // one node for the whole packet
                const float4* np = reinterpret_cast<const float4*>(sc.triNodes + cur);
                const float4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
                const float ax = w1.w * inv.x, bx = (w0.x - o.x) * inv.x;
                const float ay = w2.w * inv.y, by = (w0.y - o.y) * inv.y;
                const float az = w0.w * inv.z, bz = (w0.z - o.z) * inv.z;
                const bool px = inv.x >= 0.0f, py = inv.y >= 0.0f, pz = inv.z >= 0.0f;
                const uint32_t nxw = __float_as_uint(px ? w1.x : w2.x), fxw = __float_as_uint(px ? w2.x : w1.x);
                const uint32_t nyw = __float_as_uint(py ? w1.y : w2.y), fyw = __float_as_uint(py ? w2.y : w1.y);
                const uint32_t nzw = __float_as_uint(pz ? w1.z : w2.z), fzw = __float_as_uint(pz ? w2.z : w1.z);
                int rf[4] = { __float_as_int(w3.x), __float_as_int(w3.y), __float_as_int(w3.z), __float_as_int(w3.w) };
                float key[4]; // entry distance of the packet's first lane; +inf for a child no lane enters
#pragma unroll
                for (int k = 0; k < 4; ++k)
                {
                    const float nx = fmaf((float)((nxw >> (8 * k)) & 0xffu), ax, bx), fx = fmaf((float)((fxw >> (8 * k)) & 0xffu), ax, bx);
                    const float ny = fmaf((float)((nyw >> (8 * k)) & 0xffu), ay, by), fy = fmaf((float)((fyw >> (8 * k)) & 0xffu), ay, by);
                    const float nz = fmaf((float)((nzw >> (8 * k)) & 0xffu), az, bz), fz = fmaf((float)((fzw >> (8 * k)) & 0xffu), az, bz);
                    const float tnear = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
                    const float tfar = fminf(fminf(fx, fy), fminf(fz, best.t));
                    const bool hit = live && tnear <= tfar * SKH_SLAB_SLACK;
                    key[k] = __ballot(hit) != 0ull ? __shfl(tnear, firstLane) : INFINITY;
                }
                // (wave-uniform values from here on) children some lane enters, nearest first
#define SKH_PSWAP(a, b)                          \
    {                                            \
        const bool sw = key[b] < key[a];         \
        const float ka = sw ? key[b] : key[a];   \
        const float kb = sw ? key[a] : key[b];   \
        const int ra = sw ? rf[b] : rf[a];       \
        const int rb = sw ? rf[a] : rf[b];       \
        key[a] = ka, key[b] = kb;                \
        rf[a] = ra, rf[b] = rb;                  \
    }
                SKH_PSWAP(0, 1) SKH_PSWAP(2, 3) SKH_PSWAP(0, 2) SKH_PSWAP(1, 3) SKH_PSWAP(1, 2)
#undef SKH_PSWAP
                // farthest first onto the stack, the nearest becomes `cur`
#pragma unroll
                for (int k = 3; k >= 1; --k)
                    if (key[k] < INFINITY)
                    {
                        if (sp < SKH_PACKET_STACK)
                        {
                            if (lane == 0u)
                                s_pstack[sp] = rf[k];
                            ++sp;
                        }
                        else
                            *sc.overflowFlag = 1u;
                    }
                const int next = key[0] < INFINITY ? rf[0] : SKH_REF_INVALID;
                cur = next;
                if (cur != SKH_REF_INVALID)
                    continue;
            }
            else
            {
                // a leaf: every live lane tests its triangles
                const uint32_t enc = (uint32_t)~cur;
                const uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
                for (uint32_t k = 0; k < cnt; ++k)
                {
                    const float4* tp = sc.tris + 3 * (size_t)(first + k);
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    float t, u, v;
                    if (live && intersect_triangle(o, sh, tmin, best.t, mk3(a), mk3(b), mk3(c), t, u, v) && (best.found || t < best.t))
                    {
                        const uint32_t prim = __float_as_uint(a.w), hinst = __float_as_uint(b.w);
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
            // pop
            if (sp == 0)
                break;
            --sp;
            __builtin_amdgcn_wave_barrier();
            cur = s_pstack[sp];
            cur = __builtin_amdgcn_readfirstlane(cur);
        }
        if (live)
        {
            float4* hr = hq.rec(ridx);
            hr[0] = make_float4(best.found ? best.t : -1.0f, best.u, best.v, 0.0f);
            hr[1] = make_float4(__uint_as_float(best.inst), __uint_as_float(best.prim), 0.0f, 0.0f);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

