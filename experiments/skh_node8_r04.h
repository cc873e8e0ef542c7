// SNAPSHOT (round 4; not built): the 96-byte 8-wide node with octant-ordered slots and its encoder (option `wide` = 8; measured slower in
// rounds 2 and 4: docs/LOG.md).  Its traversal branch is in skh_trace_r04_variants.h (`if constexpr (W8)`).
// 8-wide node, 96 bytes = six 16-byte words (option `wide` = 8): half the dependent fetch -> test -> stack round trips per ray of
// the 4-wide layout.  Same quantisation frame as Node4 (origin + one power-of-two cell per axis, here as the three exponent bytes),
// one byte per child per plane, explicit child references.
//   w0  o.x o.y o.z | ex | ey << 8 | ez << 16         (cell_a = 2^(e_a - 127))
//   w1  qlo_x[0..3] qlo_x[4..7] qhi_x[0..3] qhi_x[4..7]       w2: the same for y       w3: for z
//   w4  child[0..3]      w5  child[4..7]
// SLOT ORDER IS THE TRAVERSAL ORDER: children sit in the slots so that a child lying towards (-x, -y, -z) of the node's centre gets
// slot 0 and one towards (+x, +y, +z) slot 7 (bit a of the slot = which side along axis a); a ray with direction signs
// oct = (dx < 0) | (dy < 0) << 1 | (dz < 0) << 2 visits the hit slots by ascending (slot ^ oct): near side first, no sorting network
// in the traversal (Ylitie, Karras, Laine 2017, "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", sec. 3.2).
struct Node8
{
    float o[3];
    uint32_t exps;
    uint32_t qx[4], qy[4], qz[4]; // [0..1] lo planes of children 0..3 / 4..7, [2..3] hi planes
    int child[8];
};
static_assert(sizeof(Node8) == 96, "node size");

SKH_HD void encode_node8(Node8& nd, const float* nlo, const float* nhi, const float cloIn[8][3], const float chiIn[8][3],
                         const int* refsIn, int cnt)
{
    // greedy slot assignment: repeatedly the (child, free slot) pair with the largest projection of the child's centre offset on the
    // slot's diagonal (+-1, +-1, +-1)
    int slotOf[8], childAt[8];
    for (int k = 0; k < 8; ++k)
        slotOf[k] = -1, childAt[k] = -1;
    float off[8][3];
    for (int k = 0; k < cnt; ++k)
        for (int a = 0; a < 3; ++a)
            off[k][a] = (cloIn[k][a] + chiIn[k][a]) - (nlo[a] + nhi[a]);
    for (int round = 0; round < cnt; ++round)
    {
        float bestV = -3.0e38f;
        int bk = -1, bs = -1;
        for (int k = 0; k < cnt; ++k)
        {
            if (slotOf[k] >= 0)
                continue;
            for (int sl = 0; sl < 8; ++sl)
            {
                if (childAt[sl] >= 0)
                    continue;
                const float v = ((sl & 1) ? off[k][0] : -off[k][0]) + ((sl & 2) ? off[k][1] : -off[k][1]) + ((sl & 4) ? off[k][2] : -off[k][2]);
                if (v > bestV)
                    bestV = v, bk = k, bs = sl;
            }
        }
        slotOf[bk] = bs;
        childAt[bs] = bk;
    }
    float m = 0.0f;
    for (int a = 0; a < 3; ++a)
        m = fmaxf(m, fmaxf(fabsf(nlo[a]), fabsf(nhi[a])));
    const float margin = m * 0x1p-20f + 1e-30f;
    nd.exps = 0;
    for (int a = 0; a < 3; ++a)
    {
        const float o = nlo[a] - margin;
        const float ext = (nhi[a] + margin) - o;
        int e = 0;
        (void)frexpf(fmaxf(ext, 1e-37f) / 255.0f, &e);
        int biased = e + 127;
        biased = biased < 1 ? 1 : (biased > 254 ? 254 : biased);
        const float inv_cell = ldexpf(1.0f, 127 - biased);
        nd.o[a] = o;
        nd.exps |= (uint32_t)biased << (8 * a);
        uint32_t w[4] = { 0, 0, 0, 0 };
        for (int sl = 0; sl < 8; ++sl)
        {
            uint32_t ql = 255u, qh = 0u; // empty slot: never overlaps
            const int k = childAt[sl];
            if (k >= 0)
            {
                const float fl = floorf((cloIn[k][a] - margin - o) * inv_cell);
                const float fh = ceilf((chiIn[k][a] + margin - o) * inv_cell);
                ql = (uint32_t)fminf(fmaxf(fl, 0.0f), 255.0f);
                qh = (uint32_t)fminf(fmaxf(fh, 0.0f), 255.0f);
            }
            w[sl >> 2] |= ql << (8 * (sl & 3));
            w[2 + (sl >> 2)] |= qh << (8 * (sl & 3));
        }
        uint32_t* dst = a == 0 ? nd.qx : (a == 1 ? nd.qy : nd.qz);
        for (int j = 0; j < 4; ++j)
            dst[j] = w[j];
    }
    for (int sl = 0; sl < 8; ++sl)
        nd.child[sl] = childAt[sl] >= 0 ? refsIn[childAt[sl]] : SKH_REF_INVALID;
}
