// strelka_hip -- BVH construction on the GPU (gfx950).
//
// Replaces the optixAccelBuild call sites of the reference (src/render/optix/OptixRender.cpp:300,366,487):
// one build over ALL triangles of all meshes (per-mesh BLAS = the cluster / radix-tree node covering exactly that mesh's
// key range), one over all curve sub-segments; the TLAS over the instances is built on the host (strelka_hip.hip).
//
// Pipeline (all kernels hand-written; no rocPRIM/hipCUB):
//   k_tri_boxes / k_seg_boxes   per-primitive boxes (curve segments: Bezier hull of a parameter sub-range)
//   k_group_bounds              per-group (mesh / curve set) AABB: wave reduction + ordered-uint atomics
//   k_morton                    30-bit Morton code of the primitive centroid inside its group's box; key = group:morton
//   radix sort                  8 bits per pass, stable: k_rs_hist -> k_rs_scan -> k_rs_scatter (wave-ballot multisplit)
//   builder (a) PLOC            k_ploc_init / _nn / _merge / _compact: agglomerative clustering over the Morton order
//           (b) Karras 2012     k_karras radix tree + k_refit (second-arriver pattern with agent-scope fences)
//   k_collapse                  level-by-level collapse of the binary tree into 4-wide 64-byte nodes with quantised child
//                               boxes (Node4), subtrees of <= leafMax primitives become leaves, leaf order assigned top-down
//   k_group_roots               each group's root reference
//   k_gather_tris / k_gather_segs   leaf-order primitive records (+ bounding cylinders of the curve sub-segments)
// Child reference: >= 0 internal node index; < 0 leaf: ~ref = (first << 3) | (count - 1); INT_MIN is the instance-exit sentinel.
#pragma once
#include "skh_device.h"

namespace skh
{

#define SKH_REF_INVALID 0x7fffffff
#define SKH_REF_SENTINEL ((int)0x80000000)
#define SKH_PRIM_DIRECT 0x80000000u // primitive word of a hit on a BAKED triangle: the low 31 bits index its shading record (k_gather_tris)

struct Node64
{
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int left, right, pad0, pad1;
};
static_assert(sizeof(Node64) == 64, "node size");

// 4-wide node with quantised child boxes, 64 bytes: ONE 64-byte fetch per visited node yields four child boxes,
// halving the number of dependent memory round trips per ray relative to the binary layout (the traversal kernel is
// latency bound, not bandwidth bound: DESIGN.md section 4).
//   o[3]    node-box origin (min corner minus a safety margin)
//   cell*   cell size per axis, a power of two 2^(e-127) chosen so that 255 cells cover the extent, stored as the float itself
//           (cellz in the origin's fourth word, cellx / celly in the spare words of the two plane rows): the traversal multiplies
//           them straight into the reciprocal direction
//   qlo/qhi one byte per child per axis (byte c of word a = child c, axis a): child box = o + q * cell, with qlo rounded
//           down and qhi rounded up, so the decoded box always contains the exact (inflated) child box
//   child   >= 0 node index, < 0 leaf ~((first << 3) | (count - 1)), SKH_REF_INVALID for an empty slot (qlo 255 > qhi 0)
struct Node4
{
    float o[3];
    float cellz;
    uint32_t qlo[3];
    float cellx;
    uint32_t qhi[3];
    float celly;
    int child[4];
};
static_assert(sizeof(Node4) == 64, "node size");

#define SKH_HD __host__ __device__ inline
SKH_HD void encode_node4(Node4& nd, const float* nlo, const float* nhi, const float cloIn[4][3], const float chiIn[4][3],
                         const int* refsIn, int cnt)
{
    // Children go into the slots by ascending surface area.  The closest-hit traversal orders them by entry distance (children entered at the
    // same distance -- the ray starts inside both -- in slot order: smaller first; the reverse measured +3.5 % closest-hit time); the any-hit
    // traversal visits slot 3 first, i.e. the child most likely to hold an occluder.  (Round 5 tried ordering by the number of primitives
    // below instead: closest-hit -1.5 %, any-hit +4 ... +12 % on the three kitchens.  Area stays.)
    float clo[4][3], chi[4][3];
    int refs[4];
    {
        int ord[4] = { 0, 1, 2, 3 };
        float area[4];
        for (int c = 0; c < 4; ++c)
        {
            const float ex = chiIn[c < cnt ? c : 0][0] - cloIn[c < cnt ? c : 0][0], ey = chiIn[c < cnt ? c : 0][1] - cloIn[c < cnt ? c : 0][1],
                        ez = chiIn[c < cnt ? c : 0][2] - cloIn[c < cnt ? c : 0][2];
            area[c] = c < cnt ? ex * ey + ey * ez + ez * ex : 3.0e38f; // (empty slots stay behind the used ones)
        }
        for (int i = 1; i < cnt; ++i) // insertion sort of the used slots
            for (int j = i; j > 0 && area[ord[j]] < area[ord[j - 1]]; --j)
            {
                const int t = ord[j];
                ord[j] = ord[j - 1];
                ord[j - 1] = t;
            }
        for (int c = 0; c < 4; ++c)
        {
            const int sIdx = c < cnt ? ord[c] : 0;
            for (int a = 0; a < 3; ++a)
                clo[c][a] = cloIn[sIdx][a], chi[c][a] = chiIn[sIdx][a];
            refs[c] = c < cnt ? refsIn[sIdx] : SKH_REF_INVALID;
        }
    }
    float m = 0.0f;
    for (int a = 0; a < 3; ++a)
        m = fmaxf(m, fmaxf(fabsf(nlo[a]), fabsf(nhi[a])));
    const float margin = m * 0x1p-20f + 1e-30f;
    nd.cellx = nd.celly = nd.cellz = 0.0f;
    for (int a = 0; a < 3; ++a)
    {
        const float o = nlo[a] - margin;
        const float ext = (nhi[a] + margin) - o;
        int e = 0;
        (void)frexpf(fmaxf(ext, 1e-37f) / 255.0f, &e); // ext / 255 = f * 2^e, f in [0.5, 1)  =>  255 * 2^e > ext
        int biased = e + 127;
        biased = biased < 1 ? 1 : (biased > 254 ? 254 : biased);
        const float inv_cell = ldexpf(1.0f, 127 - biased);
        nd.o[a] = o;
        (a == 0 ? nd.cellx : (a == 1 ? nd.celly : nd.cellz)) = ldexpf(1.0f, biased - 127);
        uint32_t wl = 0, wh = 0;
        for (int c = 0; c < 4; ++c)
        {
            uint32_t ql = 255u, qh = 0u; // empty slot: never overlaps
            if (c < cnt)
            {
                const float fl = floorf((clo[c][a] - margin - o) * inv_cell);
                const float fh = ceilf((chi[c][a] + margin - o) * inv_cell);
                ql = (uint32_t)fminf(fmaxf(fl, 0.0f), 255.0f);
                qh = (uint32_t)fminf(fmaxf(fh, 0.0f), 255.0f);
            }
            wl |= ql << (8 * c);
            wh |= qh << (8 * c);
        }
        nd.qlo[a] = wl;
        nd.qhi[a] = wh;
    }
    for (int c = 0; c < 4; ++c)
        nd.child[c] = c < cnt ? refs[c] : SKH_REF_INVALID;
}

SKH_DI int make_leaf_ref(uint32_t first, uint32_t count)
{
    return ~(int)((first << 3) | (count - 1u));
}
SKH_DI uint32_t ordered_from_float(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
SKH_DI float float_from_ordered(uint32_t u)
{
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}
SKH_DI uint32_t expand_bits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

SKH_DI uint64_t expand_bits21(uint64_t v) // bit k -> bit 3k
{
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// ---- per-primitive boxes -------------------------------------------------------------------------------
// triangles: box of the three object-space positions; group = mesh id
__global__ void k_tri_boxes(const uint8_t* __restrict__ verts /*32 B stride*/, const uint32_t* __restrict__ indices,
                            const uint4* __restrict__ meshes /*index_offset,index_count,vertex_offset,vertex_count*/,
                            const uint32_t* __restrict__ triMesh, const uint32_t* __restrict__ triLocal, uint32_t n,
                            float4* __restrict__ boxLo, float4* __restrict__ boxHi, uint32_t* __restrict__ grp)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t m = triMesh[i], t = triLocal[i];
    const uint4 me = meshes[m];
    v3 lo = mk3(INFINITY), hi = mk3(-INFINITY);
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        const uint32_t vi = me.z + indices[me.x + 3 * t + k];
        const float* p = reinterpret_cast<const float*>(verts + (size_t)vi * 32);
        lo = mk3(fminf(lo.x, p[0]), fminf(lo.y, p[1]), fminf(lo.z, p[2]));
        hi = mk3(fmaxf(hi.x, p[0]), fmaxf(hi.y, p[1]), fmaxf(hi.z, p[2]));
    }
    boxLo[i] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    boxHi[i] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    grp[i] = m;
}
// curve segments: box of the 4 B-spline control points (convex hull property) grown by the largest radius
// ---- curve segments, optionally cut into K parameter sub-ranges ("sub-segments") --------------------------------
// A long thin diagonal hair fills a tiny part of its box; K sub-ranges give K boxes that hug the curve.  Sub-primitive p of a
// curve set = (segment p / K, sub-range p % K).  Every sub-leaf runs the SAME full-segment intersector (so (t, u) are the bits
// the unsplit segment gives) and keeps the hit only when u falls into its own sub-range: the nearest hit is found by exactly
// the leaf that owns its u.  For that leaf's box to contain the hit point P: P lies on the sphere of some u* near u with
// |C(u*) - C(u)| <= r for |dr/ds| <= 1, i.e. within 2 r of C(u); the box is the hull of the sub-curve's Bezier points over the
// sub-range widened by SKH_SUBSEG_PAD on both sides, grown by 2 r_max (and rounding slack).
#define SKH_SUBSEG_PAD 0.02f
SKH_DI float4 lerp4(const float4& a, const float4& b, float t)
{
    return make_float4(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t, a.w + (b.w - a.w) * t);
}
SKH_DI float4 bezier_blossom(const float4* b, float t0, float t1, float t2)
{
    const float4 l0 = lerp4(b[0], b[1], t0), l1 = lerp4(b[1], b[2], t0), l2 = lerp4(b[2], b[3], t0);
    const float4 m0 = lerp4(l0, l1, t1), m1 = lerp4(l1, l2, t1);
    return lerp4(m0, m1, t2);
}
// Bezier control points (xyz, radius) of the part [u0, u1] of the uniform cubic B-spline segment q[0..3]
SKH_DI void subcurve_bezier(const float4* q, float u0, float u1, float4* c)
{
    float4 b[4];
    b[0] = make_float4((q[0].x + 4.0f * q[1].x + q[2].x) / 6.0f, (q[0].y + 4.0f * q[1].y + q[2].y) / 6.0f, (q[0].z + 4.0f * q[1].z + q[2].z) / 6.0f,
                       (q[0].w + 4.0f * q[1].w + q[2].w) / 6.0f);
    b[1] = make_float4((2.0f * q[1].x + q[2].x) / 3.0f, (2.0f * q[1].y + q[2].y) / 3.0f, (2.0f * q[1].z + q[2].z) / 3.0f, (2.0f * q[1].w + q[2].w) / 3.0f);
    b[2] = make_float4((q[1].x + 2.0f * q[2].x) / 3.0f, (q[1].y + 2.0f * q[2].y) / 3.0f, (q[1].z + 2.0f * q[2].z) / 3.0f, (q[1].w + 2.0f * q[2].w) / 3.0f);
    b[3] = make_float4((q[1].x + 4.0f * q[2].x + q[3].x) / 6.0f, (q[1].y + 4.0f * q[2].y + q[3].y) / 6.0f, (q[1].z + 4.0f * q[2].z + q[3].z) / 6.0f,
                       (q[1].w + 4.0f * q[2].w + q[3].w) / 6.0f);
    c[0] = bezier_blossom(b, u0, u0, u0);
    c[1] = bezier_blossom(b, u0, u0, u1);
    c[2] = bezier_blossom(b, u0, u1, u1);
    c[3] = bezier_blossom(b, u1, u1, u1);
}
SKH_DI void subseg_range(uint32_t sub, uint32_t K, float& u0, float& u1)
{
    u0 = K > 1u ? fmaxf((float)sub / (float)K - SKH_SUBSEG_PAD, 0.0f) : 0.0f;
    u1 = K > 1u ? fminf((float)(sub + 1u) / (float)K + SKH_SUBSEG_PAD, 1.0f) : 1.0f;
}
// box of sub-range `sub` of K of the segment with control points q (xyz, radius): hull of the (padded) sub-curve's Bezier points, grown (see above)
SKH_DI void subseg_box(const float4* q, uint32_t sub, uint32_t K, float rmax, float cmax, v3& lo, v3& hi)
{
    float4 c[4];
    float u0, u1;
    subseg_range(sub, K, u0, u1);
    subcurve_bezier(q, u0, u1, c);
    lo = mk3(INFINITY), hi = mk3(-INFINITY);
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        lo = mk3(fminf(lo.x, c[k].x), fminf(lo.y, c[k].y), fminf(lo.z, c[k].z));
        hi = mk3(fmaxf(hi.x, c[k].x), fmaxf(hi.y, c[k].y), fmaxf(hi.z, c[k].z));
    }
    const float m = (K > 1u ? 2.0f : 1.0f) * rmax * 1.01f + cmax * 4e-6f;
    lo = mk3(lo.x - m, lo.y - m, lo.z - m);
    hi = mk3(hi.x + m, hi.y + m, hi.z + m);
}
SKH_DI void load_segment(const float* __restrict__ points, const float* __restrict__ radii, uint32_t s, float4* q, float& rmax, float& cmax)
{
    rmax = 0.0f, cmax = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        const float* p = points + 3 * (size_t)(s + k);
        q[k] = make_float4(p[0], p[1], p[2], radii[s + k]);
        rmax = fmaxf(rmax, fabsf(q[k].w));
        cmax = fmaxf(cmax, fmaxf(fabsf(p[0]), fmaxf(fabsf(p[1]), fabsf(p[2]))));
    }
}
__global__ void k_seg_boxes(const float* __restrict__ points, const float* __restrict__ radii,
                            const uint32_t* __restrict__ segStart, const uint32_t* __restrict__ segCurve, uint32_t n /*segments x K*/,
                            uint32_t K, float4* __restrict__ boxLo, float4* __restrict__ boxHi, uint32_t* __restrict__ grp)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t seg = i / K, sub = i - seg * K;
    float4 q[4];
    float rmax, cmax;
    load_segment(points, radii, segStart[seg], q, rmax, cmax);
    v3 lo, hi;
    subseg_box(q, sub, K, rmax, cmax, lo, hi);
    boxLo[i] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    boxHi[i] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    grp[i] = segCurve[seg];
}

// ---- segment nodes (round 6) -----------------------------------------------------------------------------------------------------
// The curve tree is built over whole SEGMENTS (one per leaf); under every leaf sits a *segment node*: a Node4 whose four child boxes are the
// segment's SKH_SEGNODE_K parameter sub-ranges and whose four child references all name the segment's one leaf record.  The traversal's node
// loop visits it like any node (same fetch, same four slab tests, same instructions -- no divergent code for the last level); if the ray meets
// ANY of the sub-range boxes the lane goes to the leaf pass with ONE candidate for the segment (one bounding-cylinder test, then the
// cooperative Newton block), otherwise it pops.  Against sub-ranges as primitives (curve_split = 4: 5.2 M leaves, the same segment a
// candidate once per sub-range leaf the ray enters) the tree has a quarter of the leaves, a segment is tested at most once per ray, and the
// sub-range rule of the Newton block (a sub-range leaf keeps a hit only if u is its own) is gone.  Boxes stay conservative, intersect_curve_segment
// decides: same hit records.  A reference with SKH_REF_SEGNODE set indexes the node array like any other (low 28 bits).
#define SKH_REF_SEGNODE 0x10000000
#ifndef SKH_SEG_STRIDE
#define SKH_SEG_STRIDE 8 // float4 per curve leaf record: 4 control points, 2 of bounding cylinder, 1 of ids, 1 spare (skh_kernels.h DevScene::segs)
#endif
#define SKH_SEGNODE_K 4u
// box of a whole segment for the tree above the segment nodes = the union of its sub-range boxes (so a segment node's children lie inside what its parent stores for it)
__global__ void k_seg_union_boxes(const float* __restrict__ points, const float* __restrict__ radii, const uint32_t* __restrict__ segStart,
                                  const uint32_t* __restrict__ segCurve, uint32_t nSegs, float4* __restrict__ boxLo, float4* __restrict__ boxHi,
                                  uint32_t* __restrict__ grp)
{
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= nSegs)
        return;
    float4 q[4];
    float rmax, cmax;
    load_segment(points, radii, segStart[seg], q, rmax, cmax);
    v3 lo = mk3(INFINITY), hi = mk3(-INFINITY);
    for (uint32_t sub = 0; sub < SKH_SEGNODE_K; ++sub)
    {
        v3 l, h;
        subseg_box(q, sub, SKH_SEGNODE_K, rmax, cmax, l, h);
        lo = mk3(fminf(lo.x, l.x), fminf(lo.y, l.y), fminf(lo.z, l.z));
        hi = mk3(fmaxf(hi.x, h.x), fmaxf(hi.y, h.y), fmaxf(hi.z, h.z));
    }
    boxLo[seg] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    boxHi[seg] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    grp[seg] = segCurve[seg];
}
// segment node of the segment at leaf position j (record position pos = j, or the segment's own index when the records are kept strand-major)
__global__ void k_segnode_emit(const float* __restrict__ points, const float* __restrict__ radii, const uint32_t* __restrict__ segStart,
                               const uint32_t* __restrict__ sortedVals, uint32_t nSegs, uint32_t strandMajor, Node4* __restrict__ segNodes /* nSegs, behind the tree's nodes */)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nSegs)
        return;
    const uint32_t seg = sortedVals[j];
    const uint32_t pos = strandMajor ? seg : j;
    float4 q[4];
    float rmax, cmax;
    load_segment(points, radii, segStart[seg], q, rmax, cmax);
    float clo[4][3], chi[4][3], nlo[3] = { INFINITY, INFINITY, INFINITY }, nhi[3] = { -INFINITY, -INFINITY, -INFINITY };
    int refs[4];
    for (uint32_t sub = 0; sub < SKH_SEGNODE_K; ++sub)
    {
        v3 l, h;
        subseg_box(q, sub, SKH_SEGNODE_K, rmax, cmax, l, h);
        clo[sub][0] = l.x, clo[sub][1] = l.y, clo[sub][2] = l.z;
        chi[sub][0] = h.x, chi[sub][1] = h.y, chi[sub][2] = h.z;
        for (int a = 0; a < 3; ++a)
            nlo[a] = fminf(nlo[a], clo[sub][a]), nhi[a] = fmaxf(nhi[a], chi[sub][a]);
        refs[sub] = make_leaf_ref(pos, 1u);
    }
    Node4 nd;
    encode_node4(nd, nlo, nhi, clo, chi, refs, (int)SKH_SEGNODE_K);
    segNodes[pos] = nd;
}
// child references of the tree's internal nodes (and the per-curve-set roots): a one-segment leaf -> the segment node in front of it
__global__ void k_segnode_patch(int* __restrict__ refs, uint32_t nRefs, uint32_t stride, uint32_t perBlock, const uint32_t* __restrict__ sortedVals,
                                uint32_t strandMajor, uint32_t firstSegNode)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRefs)
        return;
    int* r = refs + (size_t)(i / perBlock) * stride + (i % perBlock);
    const int v = *r;
    if (v >= 0 || v == SKH_REF_SENTINEL)
        return; // an internal node, an empty slot (SKH_REF_INVALID is positive)
    const uint32_t e = (uint32_t)~v;
    const uint32_t j = e >> 3; // (leaves hold exactly one segment in this build)
    *r = (int)(SKH_REF_SEGNODE | (firstSegNode + (strandMajor ? sortedVals[j] : j)));
}

__global__ void k_init_group_bounds(uint32_t* __restrict__ gb, uint32_t nGroups)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nGroups * 6)
        gb[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
}
// per-group (mesh / curve set) bounds with ordered-uint atomics.  Primitives arrive grouped, so a wave walks SKH_GB_RUN consecutive
// 64-primitive rows and keeps a running per-lane min / max while the rows stay inside one group; it reduces across its lanes and issues
// six atomics only when the group changes and at the end of its run.  (Round 3's version reduced every 64-primitive row by itself: with the
// default world-space bake all 23.1 M triangles belong to one or two groups, and 361 k waves x 6 returning atomics on the same six words
// ran at the ~88 per microsecond one cache line sustains: 24.6 ms of a 73.8 ms build.)
#define SKH_GB_RUN 32
SKH_DI void group_bounds_flush(uint32_t g, uint32_t v[6], uint32_t* __restrict__ gb)
{
    if (g == 0xffffffffu)
        return;
#pragma unroll
    for (int k = 0; k < 6; ++k)
    {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            const uint32_t o = (uint32_t)__shfl_xor((int)v[k], off);
            v[k] = k < 3 ? min(v[k], o) : max(v[k], o);
        }
    }
    if ((threadIdx.x & 63u) == 0)
    {
        uint32_t* p = gb + 6 * (size_t)g;
        atomicMin(p + 0, v[0]);
        atomicMin(p + 1, v[1]);
        atomicMin(p + 2, v[2]);
        atomicMax(p + 3, v[3]);
        atomicMax(p + 4, v[4]);
        atomicMax(p + 5, v[5]);
    }
}
__global__ void __launch_bounds__(256) k_group_bounds(const float4* __restrict__ boxLo, const float4* __restrict__ boxHi,
                                                      const uint32_t* __restrict__ grp, uint32_t n, uint32_t* __restrict__ gb)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const size_t first = (size_t)wave * 64u * SKH_GB_RUN;
    uint32_t run = 0xffffffffu; // the group the running bounds belong to (wave-uniform)
    uint32_t acc[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
    for (uint32_t r = 0; r < SKH_GB_RUN; ++r)
    {
        const size_t i = first + (size_t)r * 64u + lane;
        if (first + (size_t)r * 64u >= n)
            break; // (wave-uniform)
        const bool valid = i < n;
        const uint32_t g = valid ? grp[i] : 0xffffffffu;
        uint32_t v[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
        if (valid)
        {
            const float4 lo = boxLo[i], hi = boxHi[i];
            v[0] = ordered_from_float(lo.x), v[1] = ordered_from_float(lo.y), v[2] = ordered_from_float(lo.z);
            v[3] = ordered_from_float(hi.x), v[4] = ordered_from_float(hi.y), v[5] = ordered_from_float(hi.z);
        }
        const uint32_t g0 = __shfl(g, __ffsll((long long)__ballot(valid)) - 1);
        if (__all(!valid || g == g0))
        {
            if (g0 != run)
            {
                group_bounds_flush(run, acc, gb); // the row starts another group: hand the finished one over
                run = g0;
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    acc[k] = k < 3 ? 0xffffffffu : 0u;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k)
                acc[k] = k < 3 ? min(acc[k], v[k]) : max(acc[k], v[k]);
        }
        else if (valid)
        {
            // a row that straddles groups (at most one per group boundary): every lane for itself
            uint32_t* p = gb + 6 * (size_t)g;
            atomicMin(p + 0, v[0]);
            atomicMin(p + 1, v[1]);
            atomicMin(p + 2, v[2]);
            atomicMax(p + 3, v[3]);
            atomicMax(p + 4, v[4]);
            atomicMax(p + 5, v[5]);
        }
    }
    group_bounds_flush(run, acc, gb);
}
__global__ void k_decode_group_bounds(const uint32_t* __restrict__ gb, float* __restrict__ out, uint32_t nGroups)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nGroups * 6)
        out[i] = float_from_ordered(gb[i]);
}
// key = group id << (3 * mb) | Morton code of the box centre inside the group's bounds, mb bits per axis (option morton_bits:
// PLOC's neighbour search makes up for a coarse grid -- 10 to 20 bits measured equal on a 23 M-triangle world-space group)
__global__ void k_morton(const float4* __restrict__ boxLo, const float4* __restrict__ boxHi, const uint32_t* __restrict__ grp,
                         const float* __restrict__ gbounds, uint32_t n, uint32_t mb, uint64_t* __restrict__ keys,
                         uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float4 lo = boxLo[i], hi = boxHi[i];
    const uint32_t g = grp[i];
    const float* b = gbounds + 6 * (size_t)g;
    const float cx = 0.5f * (lo.x + hi.x), cy = 0.5f * (lo.y + hi.y), cz = 0.5f * (lo.z + hi.z);
    const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
    const float nx = ex > 0.0f ? (cx - b[0]) / ex : 0.0f;
    const float ny = ey > 0.0f ? (cy - b[1]) / ey : 0.0f;
    const float nz = ez > 0.0f ? (cz - b[2]) / ez : 0.0f;
    const float cells = (float)(1u << mb);
    const uint32_t qx = (uint32_t)fminf(fmaxf(nx * cells, 0.0f), cells - 1.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf(ny * cells, 0.0f), cells - 1.0f);
    const uint32_t qz = (uint32_t)fminf(fmaxf(nz * cells, 0.0f), cells - 1.0f);
    const uint64_t code = (expand_bits21(qx) << 2) | (expand_bits21(qy) << 1) | expand_bits21(qz);
    keys[i] = ((uint64_t)g << (3u * mb)) | code;
    vals[i] = i;
}

// ---- stable LSD radix sort, 8 bits per pass -----------------------------------------------------------------
#define SKH_RS_THREADS 256
#define SKH_RS_ITEMS 16 // elements per thread per block => 4096 elements per block

// n may live on the device (nPtr != nullptr): queue lengths are never read back by the host during a frame
__global__ void __launch_bounds__(SKH_RS_THREADS) k_rs_hist(const uint64_t* __restrict__ keys, uint32_t n, uint32_t shift,
                                                           uint32_t* __restrict__ hist /*[256][numBlocks]*/,
                                                           uint32_t numBlocks, const uint32_t* __restrict__ nPtr)
{
    __shared__ uint32_t h[256];
    if (nPtr)
        n = *nPtr;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (SKH_RS_THREADS * SKH_RS_ITEMS);
    for (int k = 0; k < SKH_RS_ITEMS; ++k)
    {
        const uint32_t i = base + k * SKH_RS_THREADS + threadIdx.x;
        if (i < n)
            atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * numBlocks + blockIdx.x] = h[threadIdx.x];
}
// exclusive scan of `count` uint32 in place, single workgroup of 1024 threads
__global__ void __launch_bounds__(1024) k_rs_scan(uint32_t* __restrict__ data, uint32_t count)
{
    __shared__ uint32_t s[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < count; base += 1024)
    {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < count ? data[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1)
        {
            const uint32_t t = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t incl = s[threadIdx.x];
        const uint32_t c = carry;
        if (i < count)
            data[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023)
            carry = c + incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(SKH_RS_THREADS)
    k_rs_scatter(const uint64_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn, uint64_t* __restrict__ keysOut,
                 uint32_t* __restrict__ valsOut, uint32_t n, uint32_t shift, const uint32_t* __restrict__ hist,
                 uint32_t numBlocks, const uint32_t* __restrict__ nPtr)
{
    __shared__ uint32_t digitBase[256];
    if (nPtr)
        n = *nPtr;
    if (blockIdx.x * (SKH_RS_THREADS * SKH_RS_ITEMS) >= n)
        return;
    __shared__ uint32_t waveCount[4][256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    digitBase[threadIdx.x] = hist[(size_t)threadIdx.x * numBlocks + blockIdx.x];
    for (int w = 0; w < 4; ++w)
        waveCount[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (SKH_RS_THREADS * SKH_RS_ITEMS);
    for (int k = 0; k < SKH_RS_ITEMS; ++k)
    {
        const uint32_t i = base + k * SKH_RS_THREADS + threadIdx.x;
        const bool valid = i < n;
        const uint64_t key = valid ? keysIn[i] : 0ull;
        const uint32_t val = valid ? valsIn[i] : 0u;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        // wave-level multisplit: lanes with the same digit
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b)
        {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned long long below = peers & ((1ull << lane) - 1ull);
        const uint32_t rank = __popcll(below);
        if (valid && below == 0ull)
            waveCount[wave][d] = __popcll(peers);
        __syncthreads();
        if (valid)
        {
            uint32_t off = digitBase[d] + rank;
            for (uint32_t w = 0; w < wave; ++w)
                off += waveCount[w][d];
            keysOut[off] = key;
            valsOut[off] = val;
        }
        __syncthreads();
        {
            const uint32_t t = threadIdx.x;
            digitBase[t] += waveCount[0][t] + waveCount[1][t] + waveCount[2][t] + waveCount[3][t];
            waveCount[0][t] = waveCount[1][t] = waveCount[2][t] = waveCount[3][t] = 0;
        }
        __syncthreads();
    }
}

// ---- Karras 2012 ------------------------------------------------------------------------------------------
SKH_DI int delta_keys(const uint64_t* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n)
        return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b)
        return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}
// node ids: internal i in [0, n-2]; leaf j is id (n-1)+j
__global__ void k_karras(const uint64_t* __restrict__ keys, int n, int* __restrict__ childL, int* __restrict__ childR,
                         int* __restrict__ parent, int* __restrict__ rangeF, int* __restrict__ rangeL)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1)
        return;
    const int d = (delta_keys(keys, n, i, i + 1) - delta_keys(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta_keys(keys, n, i, i - d);
    int lmax = 2;
    while (delta_keys(keys, n, i, i + lmax * d) > dmin)
        lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta_keys(keys, n, i, i + (l + t) * d) > dmin)
            l += t;
    const int j = i + l * d;
    const int dnode = delta_keys(keys, n, i, j);
    int s = 0;
    int t = l;
    do
    {
        t = (t + 1) >> 1;
        if (delta_keys(keys, n, i, i + (s + t) * d) > dnode)
            s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int L = (first == gamma) ? (n - 1) + gamma : gamma;
    const int R = (last == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
    childL[i] = L;
    childR[i] = R;
    parent[L] = i;
    parent[R] = i;
    rangeF[i] = first;
    rangeL[i] = last;
    if (i == 0)
        parent[0] = -1;
}
__global__ void k_refit(const uint32_t* __restrict__ sortedVals, const float4* __restrict__ boxLo,
                        const float4* __restrict__ boxHi, const int* __restrict__ parent, const int* __restrict__ childL,
                        const int* __restrict__ childR, uint32_t* __restrict__ flags, float4* __restrict__ nodeLo,
                        float4* __restrict__ nodeHi, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const uint32_t p = sortedVals[j];
    float4 lo = boxLo[p], hi = boxHi[p];
    int id = (n - 1) + j;
    nodeLo[id] = lo;
    nodeHi[id] = hi;
    int cur = parent[id];
    while (cur >= 0)
    {
        __threadfence(); // release our child's box (agent scope) before announcing arrival
        const uint32_t old = atomicAdd(&flags[cur], 1u);
        if (old == 0)
            return; // first arriver leaves; the second one computes the node
        __threadfence(); // acquire: the sibling's box was released before its atomic
        const int other = (childL[cur] == id) ? childR[cur] : childL[cur];
        const float4 olo = nodeLo[other];
        const float4 ohi = nodeHi[other];
        lo = make_float4(fminf(lo.x, olo.x), fminf(lo.y, olo.y), fminf(lo.z, olo.z), 0.0f);
        hi = make_float4(fmaxf(hi.x, ohi.x), fmaxf(hi.y, ohi.y), fmaxf(hi.z, ohi.z), 0.0f);
        nodeLo[cur] = lo;
        nodeHi[cur] = hi;
        id = cur;
        cur = parent[cur];
    }
}
SKH_DI void inflate_box(float4& lo, float4& hi)
{
    const float m = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(lo.y)), fmaxf(fabsf(lo.z), fabsf(hi.x))), fmaxf(fabsf(hi.y), fabsf(hi.z)));
    const float e = m * 0x1p-20f + 1e-30f;
    lo = make_float4(lo.x - e, lo.y - e, lo.z - e, 0.0f);
    hi = make_float4(hi.x + e, hi.y + e, hi.z + e, 0.0f);
}
__global__ void k_emit(const int* __restrict__ childL, const int* __restrict__ childR, const int* __restrict__ rangeF,
                       const int* __restrict__ rangeL, const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi,
                       int n, int leafMax, Node64* __restrict__ nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1)
        return;
    Node64 nd;
    int refs[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
        const int c = k == 0 ? childL[i] : childR[i];
        float4 lo = nodeLo[c], hi = nodeHi[c];
        inflate_box(lo, hi);
        float* bmin = k == 0 ? nd.lmin : nd.rmin;
        float* bmax = k == 0 ? nd.lmax : nd.rmax;
        bmin[0] = lo.x, bmin[1] = lo.y, bmin[2] = lo.z;
        bmax[0] = hi.x, bmax[1] = hi.y, bmax[2] = hi.z;
        if (c >= n - 1)
            refs[k] = make_leaf_ref((uint32_t)(c - (n - 1)), 1u);
        else
        {
            const int cnt = rangeL[c] - rangeF[c] + 1;
            refs[k] = cnt <= leafMax ? make_leaf_ref((uint32_t)rangeF[c], (uint32_t)cnt) : c;
        }
    }
    nd.left = refs[0];
    nd.right = refs[1];
    nd.pad0 = nd.pad1 = 0;
    nodes[i] = nd;
}
__global__ void k_group_roots(const int* __restrict__ rangeF, const int* __restrict__ rangeL,
                              const uint64_t* __restrict__ sortedKeys, const uint32_t* __restrict__ groupFirst,
                              const uint32_t* __restrict__ groupCount, int n, int leafMax, uint32_t keyShift, int* __restrict__ groupRoot)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1)
        return;
    const int f = rangeF[i], l = rangeL[i];
    const uint32_t g = (uint32_t)(sortedKeys[f] >> keyShift);
    if ((uint32_t)f == groupFirst[g] && (uint32_t)(l - f + 1) == groupCount[g] && (l - f + 1) > leafMax)
        groupRoot[g] = i;
}

// Collapse a binary tree into 4-wide nodes, one tree level per launch.  A work item is (binary node, output slot, first
// leaf position); its two children are opened greedily by surface area until four slots are filled; children that are
// themselves internal get an output slot from a global counter and go to the next level's queue.  Subtrees of at most
// leafMax primitives become leaves: their primitives are written to consecutive positions of `leafOrder` (the tree
// built by PLOC does not keep a subtree's primitives contiguous in Morton order, so leaf order is assigned here,
// top-down: left subtree first).  Node ids: internal [0, n-2], leaf of sorted primitive j is (n-1)+j.
struct CollapseItem
{
    int bin;
    int out;
    int first;
    int pad;
};
SKH_DI int subtree_size(const int* __restrict__ nodeSize, int c, int n)
{
    return c >= n - 1 ? 1 : nodeSize[c];
}
__global__ void k_collapse(const CollapseItem* __restrict__ qin, uint32_t nIn, const int* __restrict__ childL,
                           const int* __restrict__ childR, const int* __restrict__ nodeSize, const float4* __restrict__ nodeLo,
                           const float4* __restrict__ nodeHi, int n, int leafMax, void* __restrict__ outNodes,
                           uint32_t* __restrict__ allocCounter, CollapseItem* __restrict__ qout, uint32_t* __restrict__ nOut,
                           uint32_t* __restrict__ leafOrder, float pairRatio /* > 0: a two-primitive subtree whose box has more than pairRatio x the summed areas of its two primitives' boxes MAY be opened into two one-primitive leaves when the node has a free slot (round 6, option split_pairs) */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nIn)
        return;
    constexpr int W = 4;
    const CollapseItem it = qin[i];
    int slot[W];
    int cnt = 2;
    slot[0] = childL[it.bin];
    slot[1] = childR[it.bin];
    auto openable = [&](int c) { return c < n - 1 && nodeSize[c] > leafMax; };
    auto area = [&](int c) {
        const float4 lo = nodeLo[c], hi = nodeHi[c];
        const float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
        return ex * ey + ey * ez + ez * ex;
    };
    // a LOOSE PAIR: two primitives (both children are primitives) under a box much larger than their own boxes -- a ray that enters it mostly misses both
    auto loosePair = [&](int c) {
        return pairRatio > 0.0f && c < n - 1 && nodeSize[c] == 2 && leafMax >= 2 && childL[c] >= n - 1 && childR[c] >= n - 1 &&
               area(c) > pairRatio * (area(childL[c]) + area(childR[c]));
    };
    while (cnt < W)
    {
        int best = -1;
        float bestA = -1.0f;
        for (int k = 0; k < cnt; ++k)
            if (openable(slot[k]) || loosePair(slot[k]))
            {
                const float a = area(slot[k]);
                if (a > bestA)
                {
                    bestA = a;
                    best = k;
                }
            }
        if (best < 0)
            break;
        const int c = slot[best];
        // keep left-before-right order so that leaf positions follow a depth-first walk
        for (int k = cnt; k > best + 1; --k)
            slot[k] = slot[k - 1];
        slot[best] = childL[c];
        slot[best + 1] = childR[c];
        ++cnt;
    }
    float clo[W][3], chi[W][3];
    int refs[W];
    int first = it.first;
    // the node's internal children get CONSECUTIVE output slots (one reservation): siblings a ray visits together share 128-byte lines
    // (kitchen +1.3 ... 2.5 % against one reservation per child; starting groups on an even slot on top of that: nothing)
    uint32_t nInternal = 0;
    for (int k = 0; k < cnt; ++k)
        nInternal += openable(slot[k]) ? 1u : 0u;
    uint32_t nextOut = nInternal ? atomicAdd(allocCounter, nInternal) : 0u;
    uint32_t nextQ = nInternal ? atomicAdd(nOut, nInternal) : 0u;
    for (int k = 0; k < cnt; ++k)
    {
        const int c = slot[k];
        const float4 lo = nodeLo[c], hi = nodeHi[c];
        clo[k][0] = lo.x, clo[k][1] = lo.y, clo[k][2] = lo.z;
        chi[k][0] = hi.x, chi[k][1] = hi.y, chi[k][2] = hi.z;
        const int sz = subtree_size(nodeSize, c, n);
        if (!openable(c))
        {
            // leaf: enumerate the subtree's primitives (<= leafMax <= 8) into [first, first + sz)
            refs[k] = make_leaf_ref((uint32_t)first, (uint32_t)sz);
            int stack[8];
            int sp = 0, pos = first;
            int cur = c;
            for (;;)
            {
                if (cur >= n - 1)
                {
                    leafOrder[pos++] = (uint32_t)(cur - (n - 1));
                    if (sp == 0)
                        break;
                    cur = stack[--sp];
                }
                else
                {
                    stack[sp++] = childR[cur];
                    cur = childL[cur];
                }
            }
        }
        else
        {
            const uint32_t o = nextOut++;
            refs[k] = (int)o;
            const uint32_t q = nextQ++;
            CollapseItem ni;
            ni.bin = c;
            ni.out = (int)o;
            ni.first = first;
            ni.pad = 0;
            qout[q] = ni;
        }
        first += sz;
    }
    const float4 nl = nodeLo[it.bin], nh = nodeHi[it.bin];
    const float nlo[3] = { nl.x, nl.y, nl.z }, nhi[3] = { nh.x, nh.y, nh.z };
    Node4 nd;
    encode_node4(nd, nlo, nhi, clo, chi, refs, cnt);
    static_cast<Node4*>(outNodes)[it.out] = nd;
}
// ---- refit (round 6): the 4-wide tree's boxes recomputed from the current leaf records, topology kept -------------------------------------
// The collapse hands out node slots level by level (one launch per level, internal children take the next free slots), so the nodes of level
// L occupy one contiguous range and every child node lies in a LATER range: a refit is one launch per level, deepest first, with no atomics and
// no fences.  A leaf's box = the union of its triangles' boxes from the leaf records the traversal reads (inflated as at build time); an
// internal child's = the exact box its own refit stored in nodeBox; the node is re-encoded (origin, power-of-two cells, outward-rounded child
// planes, slots by area) exactly as the collapse encodes it.  Hit records do not depend on the hierarchy: a refitted tree returns what a
// rebuilt one returns (tests/test_gpu_parity.py::test_refit_after_a_vertex_edit_equals_a_rebuild), only the nodes visited per ray differ.
SKH_DI void ref_box(int ref, const float4* __restrict__ tris, const float4* __restrict__ nodeBox, float4& lo, float4& hi)
{
    if (ref >= 0)
    {
        lo = nodeBox[2 * (size_t)ref], hi = nodeBox[2 * (size_t)ref + 1];
        return;
    }
    const uint32_t e = (uint32_t)~ref, first = e >> 3, count = (e & 7u) + 1u;
    lo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    for (uint32_t k = 0; k < count; ++k)
    {
        float4 tl = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), th = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
#pragma unroll
        for (int v = 0; v < 3; ++v)
        {
            const float4 p = tris[3 * (size_t)(first + k) + v];
            tl = make_float4(fminf(tl.x, p.x), fminf(tl.y, p.y), fminf(tl.z, p.z), 0.0f);
            th = make_float4(fmaxf(th.x, p.x), fmaxf(th.y, p.y), fmaxf(th.z, p.z), 0.0f);
        }
        inflate_box(tl, th); // (per primitive, as k_refit does for the binary tree's leaves)
        lo = make_float4(fminf(lo.x, tl.x), fminf(lo.y, tl.y), fminf(lo.z, tl.z), 0.0f);
        hi = make_float4(fmaxf(hi.x, th.x), fmaxf(hi.y, th.y), fmaxf(hi.z, th.z), 0.0f);
    }
}
__global__ void __launch_bounds__(256) k_node4_refit_level(Node4* __restrict__ nodes, float4* __restrict__ nodeBox /* 2 per node: exact lo, hi */, uint32_t first, uint32_t count,
                                                           const float4* __restrict__ tris)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count)
        return;
    const uint32_t i = first + j;
    const Node4 old = nodes[i];
    float clo[4][3], chi[4][3], nlo[3] = { INFINITY, INFINITY, INFINITY }, nhi[3] = { -INFINITY, -INFINITY, -INFINITY };
    int refs[4];
    int cnt = 0;
    for (int s = 0; s < 4; ++s)
    {
        const int ref = old.child[s];
        if (ref == SKH_REF_INVALID)
            continue;
        float4 lo, hi;
        ref_box(ref, tris, nodeBox, lo, hi);
        clo[cnt][0] = lo.x, clo[cnt][1] = lo.y, clo[cnt][2] = lo.z;
        chi[cnt][0] = hi.x, chi[cnt][1] = hi.y, chi[cnt][2] = hi.z;
        for (int a = 0; a < 3; ++a)
            nlo[a] = fminf(nlo[a], clo[cnt][a]), nhi[a] = fmaxf(nhi[a], chi[cnt][a]);
        refs[cnt++] = ref;
    }
    Node4 nd;
    encode_node4(nd, nlo, nhi, clo, chi, refs, cnt);
    nodes[i] = nd;
    nodeBox[2 * (size_t)i] = make_float4(nlo[0], nlo[1], nlo[2], 0.0f);
    nodeBox[2 * (size_t)i + 1] = make_float4(nhi[0], nhi[1], nhi[2], 0.0f);
}
// the same for a CURVE tree: a leaf's box = the union of its sub-segments' boxes, recomputed as k_seg_boxes computes them -- the Bezier hull of the
// (padded) parameter sub-range of the segment's current control points, grown by the radius margin -- from the leaf records k_gather_segs wrote
SKH_DI void curve_ref_box(int ref, const float4* __restrict__ segs, uint32_t K, const float4* __restrict__ nodeBox, float4& lo, float4& hi)
{
    if (ref >= 0)
    {
        lo = nodeBox[2 * (size_t)ref], hi = nodeBox[2 * (size_t)ref + 1];
        return;
    }
    const uint32_t e = (uint32_t)~ref, first = e >> 3, count = (e & 7u) + 1u;
    lo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    for (uint32_t k = 0; k < count; ++k)
    {
        float4 q[4];
        float rmax = 0.0f, cmax = 0.0f;
#pragma unroll
        for (int v = 0; v < 4; ++v)
        {
            q[v] = segs[SKH_SEG_STRIDE * (size_t)(first + k) + v];
            rmax = fmaxf(rmax, fabsf(q[v].w));
            cmax = fmaxf(cmax, fmaxf(fabsf(q[v].x), fmaxf(fabsf(q[v].y), fabsf(q[v].z))));
        }
        v3 l, h;
        subseg_box(q, K > 1u ? __float_as_uint(segs[SKH_SEG_STRIDE * (size_t)(first + k) + 6].x) >> 28 : 0u, K, rmax, cmax, l, h);
        lo = make_float4(fminf(lo.x, l.x), fminf(lo.y, l.y), fminf(lo.z, l.z), 0.0f);
        hi = make_float4(fmaxf(hi.x, h.x), fmaxf(hi.y, h.y), fmaxf(hi.z, h.z), 0.0f);
    }
}
__global__ void __launch_bounds__(256) k_node4_refit_level_curves(Node4* __restrict__ nodes, float4* __restrict__ nodeBox, uint32_t first, uint32_t count,
                                                                  const float4* __restrict__ segs, uint32_t K)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count)
        return;
    const uint32_t i = first + j;
    const Node4 old = nodes[i];
    float clo[4][3], chi[4][3], nlo[3] = { INFINITY, INFINITY, INFINITY }, nhi[3] = { -INFINITY, -INFINITY, -INFINITY };
    int refs[4];
    int cnt = 0;
    for (int s = 0; s < 4; ++s)
    {
        const int ref = old.child[s];
        if (ref == SKH_REF_INVALID)
            continue;
        float4 lo, hi;
        curve_ref_box(ref, segs, K, nodeBox, lo, hi);
        clo[cnt][0] = lo.x, clo[cnt][1] = lo.y, clo[cnt][2] = lo.z;
        chi[cnt][0] = hi.x, chi[cnt][1] = hi.y, chi[cnt][2] = hi.z;
        for (int a = 0; a < 3; ++a)
            nlo[a] = fminf(nlo[a], clo[cnt][a]), nhi[a] = fmaxf(nhi[a], chi[cnt][a]);
        refs[cnt++] = ref;
    }
    Node4 nd;
    encode_node4(nd, nlo, nhi, clo, chi, refs, cnt);
    nodes[i] = nd;
    nodeBox[2 * (size_t)i] = make_float4(nlo[0], nlo[1], nlo[2], 0.0f);
    nodeBox[2 * (size_t)i + 1] = make_float4(nhi[0], nhi[1], nhi[2], 0.0f);
}
// boxes of a few references (the groups' roots: a root may be a leaf) -> 6 floats each
__global__ void k_ref_boxes(const int* __restrict__ refs, uint32_t n, const float4* __restrict__ tris, const float4* __restrict__ nodeBox, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    if (refs[i] != SKH_REF_INVALID)
        ref_box(refs[i], tris, nodeBox, lo, hi);
    out[6 * i] = lo.x, out[6 * i + 1] = lo.y, out[6 * i + 2] = lo.z, out[6 * i + 3] = hi.x, out[6 * i + 4] = hi.y, out[6 * i + 5] = hi.z;
}

__global__ void k_sizes_from_ranges(const int* __restrict__ rangeF, const int* __restrict__ rangeL, int n, int* __restrict__ nodeSize)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n - 1)
        nodeSize[i] = rangeL[i] - rangeF[i] + 1;
}
__global__ void k_iota(uint32_t* __restrict__ a, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        a[i] = i;
}
__global__ void k_permute_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ order, uint32_t n, uint32_t* __restrict__ dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] = src[order[i]];
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) over the Morton-sorted primitives ----------
// Clusters live in Morton order; every iteration each cluster looks at its R neighbours on either side (same group
// only), picks the one whose merged box has the smallest surface area, and mutually-nearest pairs merge into a new
// binary node.  The active list is compacted with a prefix sum and the loop repeats until one cluster per group is
// left.  Tree quality is close to a top-down SAH build (agglomerative clustering minimises the same area measure
// bottom-up), far better than the radix tree of the plain LBVH.
// cluster record: lo = {box min, node id as int bits}, hi = {box max, group id as uint bits}
#ifndef SKH_PLOC_RADIUS
#define SKH_PLOC_RADIUS 12
#endif
#ifndef SKH_PLOC_RADIUS_SEGS
#define SKH_PLOC_RADIUS_SEGS 8
#endif
#define SKH_PLOC_BLOCK 256
__global__ void k_ploc_init(const uint32_t* __restrict__ sortedVals, const uint64_t* __restrict__ sortedKeys,
                            const float4* __restrict__ boxLo, const float4* __restrict__ boxHi, uint32_t n,
                            float4* __restrict__ cLo, float4* __restrict__ cHi, float4* __restrict__ nodeLo,
                            float4* __restrict__ nodeHi, uint32_t keyShift)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const uint32_t p = sortedVals[j];
    const float4 lo = boxLo[p], hi = boxHi[p];
    const int id = (int)(n - 1 + j);
    cLo[j] = make_float4(lo.x, lo.y, lo.z, __int_as_float(id));
    cHi[j] = make_float4(hi.x, hi.y, hi.z, __uint_as_float((uint32_t)(sortedKeys[j] >> keyShift)));
    nodeLo[id] = lo;
    nodeHi[id] = hi;
}
// RADIUS: neighbours examined on either side.  Triangles 12, curve sub-segments 8 (measured on MI355X, kitchen / hair stand-ins:
// 4 / 8 / 12 / 16 / 24 / 48 -> 5313 / 5278 / 5355 / 5359 / 5432 / 5341 and 1549 / 1549 / 1521 / 1484 / 1440 / 1418 Mray/s -- thin
// diagonal segments merge better with their neighbours along the strand than with anything a wider search finds; 24 costs the
// kitchen WITHOUT mesh sharing the 2 % it gives the shared one); the TLAS --
// thousands of boxes of wildly different sizes that EVERY ray walks -- searches 96 either side, close to exhaustive clustering
#ifndef SKH_PLOC_RADIUS_TLAS
#define SKH_PLOC_RADIUS_TLAS 96
#endif
template <int RADIUS>
__global__ void __launch_bounds__(SKH_PLOC_BLOCK) k_ploc_nn(const float4* __restrict__ cLo, const float4* __restrict__ cHi, uint32_t m,
                                                           uint32_t* __restrict__ nn)
{
    constexpr int SKH_PLOC_RADIUS_ = RADIUS;
    __shared__ float4 sLo[SKH_PLOC_BLOCK + 2 * SKH_PLOC_RADIUS_];
    __shared__ float4 sHi[SKH_PLOC_BLOCK + 2 * SKH_PLOC_RADIUS_];
    const int base = (int)(blockIdx.x * SKH_PLOC_BLOCK) - SKH_PLOC_RADIUS_;
    for (int k = threadIdx.x; k < SKH_PLOC_BLOCK + 2 * SKH_PLOC_RADIUS_; k += SKH_PLOC_BLOCK)
    {
        const int g = base + k;
        if (g >= 0 && g < (int)m)
        {
            sLo[k] = cLo[g];
            sHi[k] = cHi[g];
        }
        else
        {
            sLo[k] = make_float4(0, 0, 0, 0);
            sHi[k] = make_float4(0, 0, 0, __uint_as_float(0xffffffffu)); // group id no real group has
        }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * SKH_PLOC_BLOCK + threadIdx.x;
    if (i >= m)
        return;
    const int me = threadIdx.x + SKH_PLOC_RADIUS_;
    const float4 lo = sLo[me], hi = sHi[me];
    const uint32_t grp = __float_as_uint(hi.w);
    float bestCost = INFINITY;
    uint32_t best = 0xffffffffu;
    for (int d = -SKH_PLOC_RADIUS_; d <= SKH_PLOC_RADIUS_; ++d)
    {
        if (d == 0)
            continue;
        const float4 olo = sLo[me + d], ohi = sHi[me + d];
        if (__float_as_uint(ohi.w) != grp)
            continue;
        const float ex = fmaxf(hi.x, ohi.x) - fminf(lo.x, olo.x);
        const float ey = fmaxf(hi.y, ohi.y) - fminf(lo.y, olo.y);
        const float ez = fmaxf(hi.z, ohi.z) - fminf(lo.z, olo.z);
        const float cost = ex * ey + ey * ez + ez * ex;
        if (cost < bestCost) // ties: the first (lowest index) candidate wins, on both sides of a pair
        {
            bestCost = cost;
            best = (uint32_t)((int)i + d);
        }
    }
    nn[i] = best;
}
__global__ void k_ploc_merge(float4* __restrict__ cLo, float4* __restrict__ cHi, const uint32_t* __restrict__ nn, uint32_t m, int n,
                             int* __restrict__ childL, int* __restrict__ childR, int* __restrict__ nodeSize,
                             float4* __restrict__ nodeLo, float4* __restrict__ nodeHi, uint32_t* __restrict__ nodeCounter,
                             uint32_t* __restrict__ flags, int* __restrict__ parent /* 2n words preset to -1, or null */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m)
        return;
    const uint32_t j = nn[i];
    uint32_t keep = 1;
    const bool pair = j != 0xffffffffu && nn[j] == i;
    // node ids: one returning atomic per wave (a 23 M-primitive build made 23 M of them on one word: 15 ms of its 38)
    const bool makes = pair && i < j;
    const unsigned long long mm = __ballot(makes);
    uint32_t idBase = 0;
    if (mm != 0ull)
    {
        const int leader = __ffsll((long long)mm) - 1;
        if ((int)(threadIdx.x & 63u) == leader)
            idBase = atomicAdd(nodeCounter, (uint32_t)__popcll(mm));
        idBase = __shfl(idBase, leader);
    }
    if (pair)
    {
        if (i < j)
        {
            const float4 lo = cLo[i], hi = cHi[i], olo = cLo[j], ohi = cHi[j];
            const int a = __float_as_int(lo.w), b = __float_as_int(olo.w);
            const int id = (int)(idBase + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u)));
            childL[id] = a;
            childR[id] = b;
            if (parent)
                parent[a] = id, parent[b] = id;
            nodeSize[id] = subtree_size(nodeSize, a, n) + subtree_size(nodeSize, b, n);
            const float4 mlo = make_float4(fminf(lo.x, olo.x), fminf(lo.y, olo.y), fminf(lo.z, olo.z), 0.0f);
            const float4 mhi = make_float4(fmaxf(hi.x, ohi.x), fmaxf(hi.y, ohi.y), fmaxf(hi.z, ohi.z), 0.0f);
            nodeLo[id] = mlo;
            nodeHi[id] = mhi;
            cLo[i] = make_float4(mlo.x, mlo.y, mlo.z, __int_as_float(id));
            cHi[i] = make_float4(mhi.x, mhi.y, mhi.z, hi.w);
        }
        else
            keep = 0;
    }
    flags[i] = keep;
}
// pos = exclusive scan of flags (k_rs_scan); writes the surviving clusters and the new count
__global__ void k_ploc_compact(const float4* __restrict__ cLo, const float4* __restrict__ cHi, const uint32_t* __restrict__ flags,
                               const uint32_t* __restrict__ pos, uint32_t m, float4* __restrict__ oLo, float4* __restrict__ oHi,
                               uint32_t* __restrict__ newCount)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m)
        return;
    if (flags[i])
    {
        oLo[pos[i]] = cLo[i];
        oHi[pos[i]] = cHi[i];
    }
    if (i == m - 1)
        *newCount = pos[i] + flags[i];
}
// after the last iteration: one cluster per non-empty group, in group order
__global__ void k_ploc_roots(const float4* __restrict__ cLo, const float4* __restrict__ cHi, uint32_t m, int* __restrict__ groupRootBin)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m)
        groupRootBin[__float_as_uint(cHi[i].w)] = __float_as_int(cLo[i].w);
}

// ---- parallel reinsertion (after Meister & Bittner 2018, "Parallel reinsertion for bounding volume hierarchy optimization") ----------
// PLOC decides every merge inside a window of the Morton order; what it gets wrong -- a long thin triangle merged early with small
// neighbours, whose box then inflates every ancestor; objects that interleave in Morton order -- shows at the upper levels, the ones every
// ray walks.  One round: every node x looks for the position in the tree where the subtree it roots would cost least (k_ri_search:
// branch-and-bound over the sibling subtrees of its ancestors), the best non-conflicting moves are applied, boxes and subtree sizes are
// recomputed (k_ri_refit).  Cost = sum of the internal nodes' box areas (the SAH with fixed leaves).
//
// Moving x from under p (sibling s, grandparent g) to become the sibling of `out`: p is re-used as the new parent of (x, out).  With the
// ancestors of x called a0 = p, a1, ... and the pivot ak = the lowest common ancestor of x and out:
//   gain = A(a0) + sum_{0<j<k} (A(aj) - A(aj')) - sum_{b on the path pivot -> out, exclusive} (A(b U x) - A(b)) - A(out U x)
// aj' = aj without x (the union of the sibling subtrees below it).  The search walks the pivot upwards; under each pivot it walks the
// subtree on the other side STACKLESSLY (parent pointers; the growth sum is added going down and subtracted coming back) and leaves a
// subtree as soon as not even a zero-cost insertion below it could beat the best gain so far.
// Conflicts: a move rewrites the child / parent words of six nodes -- x, p, s, g, out, parent(out) --, which it must own: atomicMax of
// (gain bits, x) per node, the best move wins (k_ri_claim / k_ri_own; deterministic).  The nodes in between only get new boxes from the
// refit; what they must not be is carried away by another move -- two moves that take each other's target subtree along would close a
// cycle -- so an owner steps back when a node between `out` and the pivot is the x of a BETTER move (k_ri_check).  In every would-be
// cycle the move before the best one steps back: no cycle survives.  Many moves may pass through the same upper nodes, which whole-path
// locks forbid (lab: 52 k instead of 20 k moves in the first round on the 1.6 M-triangle architectural kitchen, experiments/bvhlab).
// minSize > 1 truncates the tree: only nodes whose parent holds >= minSize primitives move, subtrees below that size are not entered --
// the upper levels of a 23 M-triangle tree at 4 % of the search work (lab: 18.6 against 18.4 nodes per bounce ray; 20.9 untouched).
SKH_DI float ri_area(const float4& lo, const float4& hi)
{
    const float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
    return ex * ey + ey * ez + ez * ex;
}
SKH_DI float ri_union_area(const float4& lo, const float4& hi, const float4& blo, const float4& bhi)
{
    const float ex = fmaxf(hi.x, bhi.x) - fminf(lo.x, blo.x), ey = fmaxf(hi.y, bhi.y) - fminf(lo.y, blo.y), ez = fmaxf(hi.z, bhi.z) - fminf(lo.z, blo.z);
    return ex * ey + ey * ez + ez * ex;
}
SKH_DI unsigned long long ri_key(float gain, int x)
{
    return ((unsigned long long)__float_as_uint(gain) << 32) | (uint32_t)x; // (gain > 0: its bits order like the value)
}
// moves[x] = {out, pivot, gain bits, -} for the nodes that want to move, whose ids are appended to cand[]
__global__ void __launch_bounds__(256) k_ri_search(const int* __restrict__ childL, const int* __restrict__ childR, const int* __restrict__ parent,
                                                   const int* __restrict__ nodeSize, const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi,
                                                   int n, int minSize, const uint8_t* __restrict__ active /* null: every node searches */, int4* __restrict__ moves,
                                                   uint32_t* __restrict__ cand, uint32_t* __restrict__ nCand)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 2 * n - 1)
        return;
    int4 result = make_int4(-1, -1, 0, -1);
    const int p0 = parent[x];
    // (sparse rounds: only last round's candidates and the nodes next to last round's moves look again -- a fifth of the search work for the same
    // tree when every third round is a full one; experiments/bvhlab: 10.35 against 10.33 nodes per bounce ray after 8 rounds)
    const bool mine = !active || active[x] != 0;
    if (mine && p0 >= 0 && parent[p0] >= 0 && nodeSize[p0] >= minSize)
    {
        const float4 ilo = nodeLo[x], ihi = nodeHi[x];
        const float Ain = ri_area(ilo, ihi);
        float base = ri_area(nodeLo[p0], nodeHi[p0]);
        float bestGain = 0.0f;
        float4 plo = make_float4(3e38f, 3e38f, 3e38f, 0.0f), phi = make_float4(-3e38f, -3e38f, -3e38f, 0.0f);
        int pivot = p0, below = x;
        for (;;)
        {
            const int l = childL[pivot];
            const int sk = l == below ? childR[pivot] : l;
            int node = sk;
            float grow = 0.0f;
            for (;;)
            {
                const float4 nlo = nodeLo[node], nhi = nodeHi[node];
                const float An = ri_area(nlo, nhi), Au = ri_union_area(nlo, nhi, ilo, ihi);
                const float g = base - grow - Au;
                if (g > bestGain && !(pivot == p0 && node == sk))
                {
                    bestGain = g;
                    result.x = node, result.y = pivot;
                }
                const float grow2 = grow + (Au - An);
                if (node < n - 1 && base - grow2 - Ain > bestGain && nodeSize[node] >= minSize)
                {
                    grow = grow2;
                    node = childL[node];
                    continue;
                }
                bool done = false;
                for (;;)
                {
                    if (node == sk)
                    {
                        done = true;
                        break;
                    }
                    const int par = parent[node];
                    if (childL[par] == node)
                    {
                        node = childR[par];
                        break;
                    }
                    const float4 qlo = nodeLo[par], qhi = nodeHi[par];
                    grow -= ri_union_area(qlo, qhi, ilo, ihi) - ri_area(qlo, qhi);
                    node = par;
                }
                if (done)
                    break;
            }
            const int up = parent[pivot];
            if (up < 0)
                break;
            const float4 slo = nodeLo[sk], shi = nodeHi[sk];
            plo = make_float4(fminf(plo.x, slo.x), fminf(plo.y, slo.y), fminf(plo.z, slo.z), 0.0f);
            phi = make_float4(fmaxf(phi.x, shi.x), fmaxf(phi.y, shi.y), fmaxf(phi.z, shi.z), 0.0f);
            if (pivot != p0)
                base += ri_area(nodeLo[pivot], nodeHi[pivot]) - ri_area(plo, phi); // a_k shrinks to a_k' once the pivot has moved past it
            below = pivot;
            pivot = up;
        }
        result.z = __float_as_int(bestGain);
    }
    if (result.x >= 0)
    {
        // the nodes that want to move, as a list: the claim / own / check / apply / mark kernels run over it (a few per cent of the nodes)
        moves[x] = result;
        const unsigned long long m = __ballot(1);
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if ((int)(threadIdx.x & 63u) == leader)
            base = atomicAdd(nCand, (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        cand[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint32_t)x;
    }
}
SKH_DI int ri_sibling(const int* __restrict__ childL, const int* __restrict__ childR, int p, int x)
{
    const int l = childL[p];
    return l == x ? childR[p] : l;
}
__global__ void __launch_bounds__(256) k_ri_claim(const int4* __restrict__ moves, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ nCand,
                                                  const int* __restrict__ childL, const int* __restrict__ childR,
                                                  const int* __restrict__ parent, unsigned long long* __restrict__ lock, uint8_t* __restrict__ activeNext)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nCand)
        return;
    const int x = (int)cand[i];
    activeNext[x] = 1; // wanted to move (it may lose): looks again next round
    const int4 m = moves[x];
    const unsigned long long k = ri_key(__int_as_float(m.z), x);
    const int p = parent[x];
    atomicMax(&lock[x], k);
    atomicMax(&lock[p], k);
    atomicMax(&lock[ri_sibling(childL, childR, p, x)], k);
    atomicMax(&lock[parent[p]], k);
    atomicMax(&lock[m.x], k);
    atomicMax(&lock[parent[m.x]], k);
}
// owners of all six announce the subtree they carry away: moving[x] = key
// (moving[] is all zero between rounds: k_ri_mark's caller clears the winners' and losers' entries through the list again)
__global__ void __launch_bounds__(256) k_ri_own(const int4* __restrict__ moves, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ nCand,
                                                const int* __restrict__ childL, const int* __restrict__ childR,
                                                const int* __restrict__ parent, const unsigned long long* __restrict__ lock,
                                                unsigned long long* __restrict__ moving)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nCand)
        return;
    const int x = (int)cand[i];
    const int4 m = moves[x];
    const unsigned long long k = ri_key(__int_as_float(m.z), x);
    const int p = parent[x];
    const bool mine = lock[x] == k && lock[p] == k && lock[ri_sibling(childL, childR, p, x)] == k && lock[parent[p]] == k && lock[m.x] == k && lock[parent[m.x]] == k;
    moving[x] = mine ? k : 0ull;
}
__global__ void __launch_bounds__(256) k_ri_check(const int4* __restrict__ moves, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ nCand,
                                                  const int* __restrict__ parent, const unsigned long long* __restrict__ moving, uint8_t* __restrict__ win,
                                                  uint32_t* __restrict__ nWin)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nCand)
        return;
    const int x = (int)cand[i];
    const unsigned long long k = moving[x];
    bool ok = k != 0ull;
    if (ok)
    {
        const int4 m = moves[x];
        for (int a = parent[m.x]; a != m.y; a = parent[a])
            if (moving[a] > k)
            {
                ok = false;
                break;
            }
    }
    win[x] = ok ? 1 : 0;
    if (ok)
        atomicAdd(nWin, 1u);
}
// (every word written here belongs to a node the move owns: no two winners touch the same word)
__global__ void __launch_bounds__(256) k_ri_apply(int4* __restrict__ moves, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ nCand,
                                                  const uint8_t* __restrict__ win, int* __restrict__ childL, int* __restrict__ childR, int* __restrict__ parent)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nCand)
        return;
    const int x = (int)cand[i];
    if (!win[x])
        return;
    const int out = moves[x].x;
    const int p = parent[x], g = parent[p];
    moves[x].w = g; // (k_ri_mark stamps the old place's ancestors from here)
    const int s = ri_sibling(childL, childR, p, x);
    if (childL[g] == p) // s takes p's place under g
        childL[g] = s;
    else
        childR[g] = s;
    parent[s] = g;
    const int po = parent[out]; // (read after the line above: out may be s)
    if (childL[po] == out) // p goes in above `out`
        childL[po] = p;
    else
        childR[po] = p;
    parent[p] = po;
    childL[p] = x;
    childR[p] = out;
    parent[out] = p;
}
// After the moves only the nodes between a moved subtree's old and new place and the root have new boxes / sizes: the winners stamp those
// paths (k_ri_mark: from g and from p upwards, stopping at a node somebody else has stamped -- that somebody goes on to the root); only those
// are recomputed.  (A full second-arriver refit of the 46 M nodes of the kitchen's tree took 147 ms per round -- its agent-scope fences, not the
// arithmetic --; the stamped paths are 1-2 % of the nodes.)
__global__ void __launch_bounds__(256) k_ri_mark(const int4* __restrict__ moves, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ nCand,
                                                 const uint8_t* __restrict__ win, const int* __restrict__ parent, const int* __restrict__ childL, const int* __restrict__ childR,
                                                 uint32_t* __restrict__ stamp, uint32_t round, uint8_t* __restrict__ activeNext)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nCand)
        return;
    const int x = (int)cand[i];
    if (!win[x])
        return;
    const int p = parent[x]; // (after k_ri_apply: p sits above `out` now; its old grandparent is the parent of its old sibling)
    // (a stamped node gets a new box: it and its children may want to move in the next round)
    for (int a = p; a >= 0; a = parent[a])
    {
        if (atomicExch(&stamp[a], round) == round)
            break;
        activeNext[a] = 1, activeNext[childL[a]] = 1, activeNext[childR[a]] = 1;
    }
    for (int a = moves[x].w /* g, recorded by k_ri_apply */; a >= 0; a = parent[a])
    {
        if (atomicExch(&stamp[a], round) == round)
            break;
        activeNext[a] = 1, activeNext[childL[a]] = 1, activeNext[childR[a]] = 1;
    }
}
SKH_DI void ri_compute(int x, const int* __restrict__ childL, const int* __restrict__ childR, float4* nodeLo, float4* nodeHi, int* nodeSize, int n)
{
    const int a = childL[x], b = childR[x];
    const float4 alo = nodeLo[a], ahi = nodeHi[a];
    const float4 blo = nodeLo[b], bhi = nodeHi[b];
    nodeLo[x] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), 0.0f);
    nodeHi[x] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.0f);
    nodeSize[x] = (a >= n - 1 ? 1 : nodeSize[a]) + (b >= n - 1 ? 1 : nodeSize[b]);
}
// Refit of the stamped nodes, level by level: k_ri_pending counts every stamped node's stamped children and lists those that have none;
// each k_ri_refit_level launch computes the nodes of its list and hands a parent on to the next launch's list when its last stamped child is
// done.  Kernel boundaries order the launches' writes and reads, so no fence is needed (the single-launch version -- second arriver goes on,
// agent-scope fences around the arrival counter -- spent 2.5-3 ms per round in ~50 dependent release / acquire pairs, each an L2 write-back).
__global__ void __launch_bounds__(256) k_ri_pending(const int* __restrict__ childL, const int* __restrict__ childR, const uint32_t* __restrict__ stamp,
                                                    uint32_t round, int n, uint32_t* __restrict__ pending, uint32_t* __restrict__ list, uint32_t* __restrict__ count)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n - 1 || stamp[x] != round)
        return;
    const int a = childL[x], b = childR[x];
    const uint32_t c = ((a < n - 1 && stamp[a] == round) ? 1u : 0u) + ((b < n - 1 && stamp[b] == round) ? 1u : 0u);
    pending[x] = c;
    if (c == 0u)
        list[atomicAdd(count, 1u)] = (uint32_t)x;
}
__global__ void __launch_bounds__(256) k_ri_refit_level(const uint32_t* __restrict__ listIn, const uint32_t* __restrict__ countIn, uint32_t* __restrict__ listOut,
                                                        uint32_t* __restrict__ countOut, const int* __restrict__ parent, const int* __restrict__ childL,
                                                        const int* __restrict__ childR, uint32_t* __restrict__ pending, float4* nodeLo, float4* nodeHi,
                                                        int* nodeSize, int n)
{
    const uint32_t m = *countIn;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
    {
        const int x = (int)listIn[i];
        ri_compute(x, childL, childR, nodeLo, nodeHi, nodeSize, n);
        const int par = parent[x];
        if (par >= 0 && atomicSub(&pending[par], 1u) == 1u) // (every ancestor of a stamped node is stamped)
            listOut[atomicAdd(countOut, 1u)] = (uint32_t)par;
    }
}
// sum of the internal nodes' half-areas -- reporting only (skh_build_info.cost_before / cost_after)
__global__ void __launch_bounds__(256) k_ri_cost(const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi, int nInternal, double* __restrict__ out)
{
    __shared__ double s[4];
    double v = 0.0;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < nInternal; x += gridDim.x * blockDim.x)
        v += (double)ri_area(nodeLo[x], nodeHi[x]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0)
        s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(out, (s[0] + s[1]) + (s[2] + s[3]));
}

// ---- multi-block exclusive scan (uint32): block scan -> scan of block totals -> add back ---------------------------
#define SKH_SCAN_BLOCK 1024
#define SKH_SCAN_ITEMS 4
__global__ void __launch_bounds__(SKH_SCAN_BLOCK) k_scan_block(uint32_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ blockSums)
{
    __shared__ uint32_t s[SKH_SCAN_BLOCK];
    const uint32_t base = blockIdx.x * (SKH_SCAN_BLOCK * SKH_SCAN_ITEMS) + threadIdx.x * SKH_SCAN_ITEMS;
    uint32_t v[SKH_SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < SKH_SCAN_ITEMS; ++k)
    {
        v[k] = base + k < n ? data[base + k] : 0u;
        sum += v[k];
    }
    s[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < SKH_SCAN_BLOCK; off <<= 1)
    {
        const uint32_t t = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = s[threadIdx.x] - sum; // exclusive prefix of this thread inside the block
#pragma unroll
    for (int k = 0; k < SKH_SCAN_ITEMS; ++k)
    {
        if (base + k < n)
            data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == SKH_SCAN_BLOCK - 1)
        blockSums[blockIdx.x] = s[threadIdx.x];
}
__global__ void __launch_bounds__(SKH_SCAN_BLOCK) k_scan_add(uint32_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ blockSums)
{
    const uint32_t add = blockSums[blockIdx.x];
    const uint32_t base = blockIdx.x * (SKH_SCAN_BLOCK * SKH_SCAN_ITEMS) + threadIdx.x * SKH_SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SKH_SCAN_ITEMS; ++k)
        if (base + k < n)
            data[base + k] += add;
}

// ---- leaf records laid out by 128-byte line (triangles; Node4 trees) ------------------------------------------------------------
// The memory system delivers random fetches by the 128-byte LINE (skh_probe_memory: 32-, 64- and 128-byte records all arrive at ~55 G
// records/s), and 48-byte triangle records at arbitrary multiples of 48 straddle a line boundary every other leaf.  After the collapse
// the leaves keep their order but get padding slots in front of them so that every leaf touches the fewest lines its size allows
// (<= 2 triangles: one).  Addressing stays `first * 48`; only `first` changes.
//   k_leaf_mark   leafCnt[first] = count for every leaf reference of the nodes and of the group roots
//   k_leaf_place  one thread per chunk of SKH_LEAF_CHUNK leaf-order positions walks its leaves; pass 1 (chunkBase == nullptr) returns the
//                 chunk's padded length (a multiple of 8 slots = 3 lines, so every chunk starts on a line), pass 2 writes remap[]
//   k_leaf_patch  rewrites the references; k_leaf_scatter: sortedVals into slot order (0xffffffff = padding slot)
#define SKH_LEAF_CHUNK 256u
SKH_DI uint32_t leaf_place(uint32_t off, uint32_t cnt, uint32_t recBytes)
{
    const uint32_t bytes = cnt * recBytes, minLines = (bytes + 127u) >> 7;
    for (;;)
    {
        const uint32_t b0 = off * recBytes;
        if (((b0 + bytes - 1u) >> 7) - (b0 >> 7) + 1u <= minLines)
            return off;
        ++off;
    }
}
__global__ void k_leaf_mark(const int* __restrict__ refs, uint32_t nRefs, uint32_t stride /*ints between the reference blocks*/, uint32_t perBlock,
                            uint8_t* __restrict__ leafCnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRefs)
        return;
    const int r = refs[(size_t)(i / perBlock) * stride + i % perBlock];
    if (r < 0 && r != SKH_REF_SENTINEL)
    {
        const uint32_t enc = (uint32_t)~r;
        leafCnt[enc >> 3] = (uint8_t)((enc & 7u) + 1u);
    }
}
__global__ void k_leaf_place(const uint8_t* __restrict__ leafCnt, uint32_t n, uint32_t recBytes, const uint32_t* __restrict__ chunkBase,
                             uint32_t* __restrict__ chunkLen, uint32_t* __restrict__ remap)
{
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t j = ch * SKH_LEAF_CHUNK;
    if (j >= n)
        return;
    const uint32_t end = min(n, j + SKH_LEAF_CHUNK);
    while (j < end && leafCnt[j] == 0u) // (the tail of a leaf that started in the chunk before)
        ++j;
    uint32_t pos = chunkBase ? chunkBase[ch] : 0u;
    while (j < end)
    {
        const uint32_t cnt = leafCnt[j];
        pos = leaf_place(pos, cnt, recBytes);
        if (remap)
            for (uint32_t k = 0; k < cnt; ++k)
                remap[j + k] = pos + k;
        pos += cnt;
        j += cnt;
    }
    if (!chunkBase)
        chunkLen[ch] = (pos + 7u) & ~7u;
}
__global__ void k_leaf_patch(int* __restrict__ refs, uint32_t nRefs, uint32_t stride, uint32_t perBlock, const uint32_t* __restrict__ remap)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRefs)
        return;
    int* p = refs + (size_t)(i / perBlock) * stride + i % perBlock;
    const int r = *p;
    if (r < 0 && r != SKH_REF_SENTINEL)
    {
        const uint32_t enc = (uint32_t)~r;
        *p = ~(int)((remap[enc >> 3] << 3) | (enc & 7u));
    }
}
__global__ void k_leaf_scatter(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ remap, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n)
        out[remap[j]] = vals[j];
}

SKH_DI uint32_t find_segment(const uint32_t* __restrict__ first, uint32_t count, uint32_t k) // largest j with first[j] <= k
{
    uint32_t lo = 0, hi = count;
    while (hi - lo > 1)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if (first[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}
// ---- baked instances ("bake_world") -------------------------------------------------------------------
// A baked mesh instance has no TLAS leaf: its triangles are carried to WORLD space once (xform_point, the arithmetic the CPU
// oracle restates) and join one extra primitive group of the triangle build, whose tree every ray walks first, in world space,
// with no instance entry.  Baked primitive k (0 <= k < W) = triangle k - wFirst[j] of instance wInst[j].
// boxes: exactly the box of the three world-space vertices that the leaf record will hold (the node encoding adds the margins)
__global__ void k_baked_tri_boxes(const uint8_t* __restrict__ instances /*64 B*/, const uint32_t* __restrict__ wInst,
                                  const uint32_t* __restrict__ wFirst, uint32_t nW, const uint8_t* __restrict__ verts,
                                  const uint32_t* __restrict__ indices, const uint4* __restrict__ meshes, uint32_t W, uint32_t offset,
                                  uint32_t group, uint32_t split /* primitives >= split: group + 1 (light proxies) */, float4* __restrict__ boxLo, float4* __restrict__ boxHi, uint32_t* __restrict__ grp)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= W)
        return;
    const uint32_t j = find_segment(wFirst, nW, k);
    const uint32_t inst = wInst[j], t = k - wFirst[j];
    const float* m = reinterpret_cast<const float*>(instances + (size_t)inst * 64);
    const uint32_t geom = reinterpret_cast<const uint32_t*>(instances + (size_t)inst * 64)[13];
    const uint4 me = meshes[geom];
    v3 lo = mk3(INFINITY), hi = mk3(-INFINITY);
#pragma unroll
    for (int c = 0; c < 3; ++c)
    {
        const uint32_t vi = me.z + indices[me.x + 3 * t + c];
        const float* p = reinterpret_cast<const float*>(verts + (size_t)vi * 32);
        const v3 w = xform_point(m, mk3(p[0], p[1], p[2]));
        lo = mk3(fminf(lo.x, w.x), fminf(lo.y, w.y), fminf(lo.z, w.z));
        hi = mk3(fmaxf(hi.x, w.x), fmaxf(hi.y, w.y), fmaxf(hi.z, w.z));
    }
    boxLo[offset + k] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    boxHi[offset + k] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    grp[offset + k] = k < split ? group : group + 1u;
}

// gather triangles into leaf order: 48 B records {v0.xyz, primId | v1.xyz, 0 | v2.xyz, 0}; primitives >= nMeshTris are baked
// ones: WORLD-space vertices, {v0, shading-record index | SKH_PRIM_DIRECT | v1, instance id | v2, 0}
__global__ void k_gather_tris(const uint8_t* __restrict__ verts, const uint32_t* __restrict__ indices,
                                    const uint4* __restrict__ meshes, const uint32_t* __restrict__ triMesh,
                                    const uint32_t* __restrict__ triLocal, const uint32_t* __restrict__ sortedVals, uint32_t n,
                                    uint32_t nMeshTris, const uint8_t* __restrict__ instances, const uint8_t* __restrict__ shadeInstances /* the copy whose
                                    light word holds a mesh instance's first shading record */, const uint32_t* __restrict__ wInst,
                                    const uint32_t* __restrict__ wFirst, uint32_t nW, uint32_t direct /* 0: baked mesh triangles keep the plain primitive index too */, float4* __restrict__ out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const uint32_t i = sortedVals[j];
    float4 r[3];
    if (i == 0xffffffffu) // padding slot of the line layout (k_leaf_place): never referenced
        r[0] = r[1] = r[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    else if (i < nMeshTris)
    {
        const uint32_t m = triMesh[i], t = triLocal[i];
        const uint4 me = meshes[m];
#pragma unroll
        for (int k = 0; k < 3; ++k)
        {
            const uint32_t vi = me.z + indices[me.x + 3 * t + k];
            const float* p = reinterpret_cast<const float*>(verts + (size_t)vi * 32);
            r[k] = make_float4(p[0], p[1], p[2], k == 0 ? __uint_as_float(t) : 0.0f);
        }
    }
    else
    {
        const uint32_t kk = i - nMeshTris;
        const uint32_t w = find_segment(wFirst, nW, kk);
        const uint32_t inst = wInst[w], t = kk - wFirst[w];
        const float* m = reinterpret_cast<const float*>(instances + (size_t)inst * 64);
        const uint32_t geom = reinterpret_cast<const uint32_t*>(instances + (size_t)inst * 64)[13];
        // (a baked LIGHT proxy keeps the plain primitive index: the `light` word of its shading record is the light's index, not a mesh base,
        // and k_shade reads no triangle record for a light hit)
        const bool isMesh = reinterpret_cast<const uint32_t*>(instances + (size_t)inst * 64)[12] == 0u;
        const uint32_t tvBase = reinterpret_cast<const uint32_t*>(shadeInstances + (size_t)inst * 64)[15];
        const uint4 me = meshes[geom];
#pragma unroll
        for (int k = 0; k < 3; ++k)
        {
            const uint32_t vi = me.z + indices[me.x + 3 * t + k];
            const float* p = reinterpret_cast<const float*>(verts + (size_t)vi * 32);
            const v3 q = xform_point(m, mk3(p[0], p[1], p[2]));
            // word 0 of a baked triangle = the index of its shading record (mesh base + t: the base sits in the shading copy of the instance
            // record, build_shading_tables) with SKH_PRIM_DIRECT set: k_shade fetches the record straight from the hit, beside the instance
            // record instead of behind it; the traversal treats the word as opaque (within an instance it orders like t, so ties break alike),
            // the raw-query output turns it back into t (k_hits_soa_to_aos)
            // (word 2: 1 = a light proxy's triangle -- any-hit queries do not see lights, merge_light_proxies)
            r[k] = make_float4(q.x, q.y, q.z, k == 0 ? __uint_as_float((isMesh && direct) ? ((tvBase + t) | SKH_PRIM_DIRECT) : t) : (k == 1 ? __uint_as_float(inst) : __uint_as_float(isMesh ? 0u : 1u)));
        }
    }
    out[3 * (size_t)j + 0] = r[0];
    out[3 * (size_t)j + 1] = r[1];
    out[3 * (size_t)j + 2] = r[2];
}

// gather curve segments into leaf order: ONE 128-byte record per sub-segment (SKH_SEG_STRIDE float4) --
// [0..3] the segment's four control points {xyz, radius} (duplicated per sub-range: one record per test); [4] [5] a conservative bounding cylinder of the (padded)
// sub-range for the cheap rejection test in front of the iterative intersector: {A.xyz, R}, {unit axis.xyz, 0}; [6] {segment index inside its curve set | sub-range << 28,
// the instance of a merged segment or ~0, 0, 0}.  The curve
// part is inside the hull of its Bezier points, so every accepted hit point lies within  max_i dist(c_i, L) + 2 r_max  of the
// line L through the part's end points (r_max for the unsplit segment, see k_seg_boxes); a degenerate chord gives axis = 0,
// which switches the test off.
__global__ void k_gather_segs(const float* __restrict__ points, const float* __restrict__ radii,
                              const uint32_t* __restrict__ segStart, const uint32_t* __restrict__ segLocal, const uint32_t* __restrict__ segInstOf /* per build primitive: the instance of a merged segment, ~0 otherwise */,
                              const uint32_t* __restrict__ sortedVals, uint32_t n /*segments x K*/, uint32_t K, float4* __restrict__ out, uint32_t strandMajor /* (segment-node build, K = 1) records at the segment's own index: consecutive segments of a strand adjacent in memory */)
{
    const uint32_t jj = blockIdx.x * blockDim.x + threadIdx.x;
    if (jj >= n)
        return;
    const uint32_t i = sortedVals[jj];
    const uint32_t j = strandMajor ? i : jj;
    const uint32_t seg = i / K, sub = i - seg * K;
    const uint32_t s = segStart[seg];
    float4 q[4], c[4];
    float rmax = 0.0f, cmax = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        const float* p = points + 3 * (size_t)(s + k);
        q[k] = make_float4(p[0], p[1], p[2], radii[s + k]);
        out[SKH_SEG_STRIDE * (size_t)j + k] = q[k];
        rmax = fmaxf(rmax, fabsf(q[k].w));
        cmax = fmaxf(cmax, fmaxf(fabsf(p[0]), fmaxf(fabsf(p[1]), fabsf(p[2]))));
    }
    out[SKH_SEG_STRIDE * (size_t)j + 6] = make_float4(__uint_as_float(segLocal[seg] | (sub << 28)), __uint_as_float(segInstOf[seg]), 0.0f, 0.0f);
    out[SKH_SEG_STRIDE * (size_t)j + 7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float u0, u1;
    subseg_range(sub, K, u0, u1);
    subcurve_bezier(q, u0, u1, c);
    const float ax = c[0].x, ay = c[0].y, az = c[0].z;
    float ux = c[3].x - ax, uy = c[3].y - ay, uz = c[3].z - az;
    const float len = sqrtf(ux * ux + uy * uy + uz * uz);
    float dmax = 0.0f;
    if (len > 1e-20f)
    {
        ux /= len, uy /= len, uz /= len;
#pragma unroll
        for (int k = 1; k < 3; ++k)
        {
            const float vx = c[k].x - ax, vy = c[k].y - ay, vz = c[k].z - az;
            const float t = vx * ux + vy * uy + vz * uz;
            const float px = vx - t * ux, py = vy - t * uy, pz = vz - t * uz;
            dmax = fmaxf(dmax, sqrtf(px * px + py * py + pz * pz));
        }
    }
    else
        ux = uy = uz = 0.0f;
    const float R = (dmax + (K > 1u ? 2.0f : 1.0f) * rmax) * 1.001f + cmax * 4e-6f + 1e-30f;
    out[SKH_SEG_STRIDE * (size_t)j + 4] = make_float4(ax, ay, az, R);
    out[SKH_SEG_STRIDE * (size_t)j + 5] = make_float4(ux, uy, uz, 0.0f);
}

} // namespace skh
