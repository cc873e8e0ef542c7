// bvhlab -- CPU laboratory for the tree-quality pass of round 5 (NOT part of the product, not built by build()).
//
// Rebuilds the GPU builder's binary tree on the host (Morton sort, PLOC with the same radius / tie rule as skh_bvh.h), applies candidate
// optimisations (parallel reinsertion emulated batch by batch exactly as the GPU kernels do it, triangle pre-splitting, collapse rules) and
// prices them with a traversal simulator over the same 4-wide nodes / leaves of <= 2 the kernels walk: node visits and triangle tests per
// camera ray, per bounce ray (cosine-distributed from the camera hits) and per any-hit shadow ray.  It exists so that an idea costs a
// minute on 8 host cores instead of a GPU lease; what survives here is written in HIP and measured on the GPU.
//
//   g++ -O3 -march=native -fopenmp -std=c++17 -o /tmp/bvhlab experiments/bvhlab/bvhlab.cpp
//   /tmp/bvhlab /tmp/lab_arch [reinsert=K] [minsize=S] [split=F] [radius=R] [collapse=sah] [rays=N]
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>
#include <omp.h>

struct Box
{
    float lo[3], hi[3];
};
static inline Box empty_box()
{
    return Box{ { 3e38f, 3e38f, 3e38f }, { -3e38f, -3e38f, -3e38f } };
}
static inline Box merge(const Box& a, const Box& b)
{
    Box r;
    for (int k = 0; k < 3; ++k)
        r.lo[k] = std::min(a.lo[k], b.lo[k]), r.hi[k] = std::max(a.hi[k], b.hi[k]);
    return r;
}
static inline float area(const Box& b)
{
    const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
    return ex * ey + ey * ez + ez * ex;
}
static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Tree
{
    int n = 0; // primitives (references); nodes: internal [0, n-2], leaf of sorted reference j = (n-1)+j
    std::vector<int> L, R, parent, size;
    std::vector<Box> box; // 2n-1
    int root = -1;
    std::vector<uint32_t> prim; // sorted reference j -> triangle index
};

static uint32_t expand10(uint32_t v)
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

static void build_ploc(Tree& t, const std::vector<Box>& pb, const std::vector<uint32_t>& primOf, int radius)
{
    const int n = (int)pb.size();
    t.n = n;
    Box sb = empty_box();
    for (const Box& b : pb)
        sb = merge(sb, b);
    std::vector<uint64_t> key(n);
#pragma omp parallel for
    for (int i = 0; i < n; ++i)
    {
        uint32_t q[3];
        for (int k = 0; k < 3; ++k)
        {
            const float c = 0.5f * (pb[i].lo[k] + pb[i].hi[k]);
            const float e = sb.hi[k] - sb.lo[k];
            float f = e > 0 ? (c - sb.lo[k]) / e : 0.0f;
            q[k] = (uint32_t)std::min(1023.0f, std::max(0.0f, f * 1024.0f));
        }
        key[i] = ((uint64_t)(expand10(q[0]) | (expand10(q[1]) << 1) | (expand10(q[2]) << 2)) << 32) | (uint32_t)i;
    }
    std::sort(key.begin(), key.end());
    t.prim.resize(n);
    t.L.assign(n, -1), t.R.assign(n, -1), t.size.assign(n, 0), t.parent.assign(2 * n, -1);
    t.box.resize(2 * n);
    std::vector<Box> cb(n), cb2(n);
    std::vector<int> cid(n), cid2(n), nn(n);
    for (int j = 0; j < n; ++j)
    {
        const uint32_t i = (uint32_t)key[j];
        t.prim[j] = primOf[i];
        cb[j] = pb[i];
        cid[j] = n - 1 + j;
        t.box[n - 1 + j] = pb[i];
    }
    int m = n, next = 0;
    while (m > 1)
    {
#pragma omp parallel for schedule(static, 4096)
        for (int i = 0; i < m; ++i)
        {
            float best = INFINITY;
            int bi = -1;
            for (int d = -radius; d <= radius; ++d)
            {
                const int j = i + d;
                if (d == 0 || j < 0 || j >= m)
                    continue;
                const float c = area(merge(cb[i], cb[j]));
                if (c < best)
                    best = c, bi = j;
            }
            nn[i] = bi;
        }
        int m2 = 0;
        for (int i = 0; i < m; ++i)
        {
            const int j = nn[i];
            if (j >= 0 && nn[j] == i)
            {
                if (i < j)
                {
                    const int id = next++;
                    t.L[id] = cid[i], t.R[id] = cid[j];
                    t.parent[cid[i]] = id, t.parent[cid[j]] = id;
                    t.box[id] = merge(cb[i], cb[j]);
                    cb2[m2] = t.box[id], cid2[m2] = id, ++m2;
                }
            }
            else
                cb2[m2] = cb[i], cid2[m2] = cid[i], ++m2;
        }
        cb.swap(cb2), cid.swap(cid2);
        m = m2;
    }
    t.root = cid[0];
    t.parent[t.root] = -1;
}

// boxes and subtree sizes bottom-up
static void refit(Tree& t)
{
    const int n = t.n;
    std::vector<int> order;
    order.reserve(n);
    std::vector<int> st{ t.root };
    while (!st.empty())
    {
        const int x = st.back();
        st.pop_back();
        if (x >= n - 1)
            continue;
        order.push_back(x);
        st.push_back(t.L[x]);
        st.push_back(t.R[x]);
    }
    for (int k = (int)order.size() - 1; k >= 0; --k)
    {
        const int x = order[k];
        const int a = t.L[x], b = t.R[x];
        t.box[x] = merge(t.box[a], t.box[b]);
        t.size[x] = (a >= n - 1 ? 1 : t.size[a]) + (b >= n - 1 ? 1 : t.size[b]);
        t.parent[a] = x, t.parent[b] = x;
    }
}
static double sah_internal(const Tree& t)
{
    double s = 0;
    for (int x = 0; x < t.n - 1; ++x)
        s += area(t.box[x]);
    return s / area(t.box[t.root]);
}

// ---- parallel reinsertion (Meister & Bittner 2018), one batch: search on the frozen tree, path locks by gain, apply, refit ----
struct Move
{
    int out, pivot;
    float gain;
};
static inline int sibling(const Tree& t, int x)
{
    const int p = t.parent[x];
    return t.L[p] == x ? t.R[p] : t.L[p];
}
// best new position for subtree `in`.  Nodes whose subtree holds fewer than minSize primitives are not entered (and are not moved).
static bool ancestors = false;
static Move find_best(const Tree& t, int in, int minSize, long* visits)
{
    const int n = t.n;
    Move mv{ -1, -1, 0.0f };
    const int p0 = t.parent[in];
    if (p0 < 0 || t.parent[p0] < 0)
        return mv; // root, or a child of the root (the paper leaves those where they are)
    const Box bin = t.box[in];
    const float Ain = area(bin);
    float base = area(t.box[p0]); // a0 goes away
    Box pivotBox = empty_box(); // a_{k-1}' : the path node below the pivot, without `in`
    int pivot = p0, below = in;
    long v = 0;
    for (;;)
    {
        // the subtree hanging off the pivot on the other side
        const int sk = t.L[pivot] == below ? t.R[pivot] : t.L[pivot];
        // DFS over subtree sk with an explicit stack of (node, growth so far)
        struct E
        {
            int node;
            float grow;
        };
        E stack[128];
        int sp = 0;
        stack[sp++] = E{ sk, 0.0f };
        while (sp)
        {
            const E e = stack[--sp];
            ++v;
            const Box u = merge(t.box[e.node], bin);
            const float Au = area(u);
            const float g = base - e.grow - Au;
            if (g > mv.gain && !(pivot == p0 && e.node == sk))
                mv.gain = g, mv.out = e.node, mv.pivot = pivot;
            if (e.node < n - 1 && t.size[e.node] >= minSize)
            {
                const float grow2 = e.grow + (Au - area(t.box[e.node]));
                if (base - grow2 - Ain > mv.gain && sp + 2 <= 128)
                {
                    stack[sp++] = E{ t.L[e.node], grow2 };
                    stack[sp++] = E{ t.R[e.node], grow2 };
                }
            }
        }
        // pivot one level up
        const int up = t.parent[pivot];
        if (up < 0)
            break;
        pivotBox = merge(pivotBox, t.box[sk]);
        if (pivot != p0)
        {
            // `in` as the new sibling of its own ancestor a_k (which has shrunk to a_k' = pivotBox): the new node has a_k's old box
            const float g = base - area(pivotBox);
            if (ancestors && g > mv.gain)
                mv.gain = g, mv.out = pivot, mv.pivot = up;
            base += area(t.box[pivot]) - area(pivotBox); // a_k shrinks to a_k' once the pivot has moved past it
        }
        below = pivot;
        pivot = up;
    }
    if (visits)
        *visits += v;
    return mv;
}

static std::vector<char> g_active; // (sparse=1) nodes searched in this round: last round's candidates and the nodes next to last round's moves
static bool g_sparse = false;
static int reinsertion_batch(Tree& t, int minSize, int stride, int phase, double* gainSum, long* visitsOut, bool pathLocks, long* candOut)
{
    const int n = t.n, N = 2 * n - 1;
    std::vector<Move> mv(N, Move{ -1, -1, 0.0f });
    long visits = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : visits)
    for (int x = 0; x < N; ++x)
    {
        if (stride > 1 && (x % stride) != phase)
            continue;
        if (g_sparse && !g_active.empty() && !g_active[x])
            continue;
        const int sz = x >= n - 1 ? 1 : t.size[x];
        if (sz < minSize && !(t.parent[x] >= 0 && t.size[t.parent[x]] >= minSize))
            continue; // (below the truncation: neither it nor its parent is a node of the truncated tree)
        mv[x] = find_best(t, x, minSize, &visits);
    }
    // Conflict resolution.  A move rewrites the child / parent words of six nodes: in, its parent p, sibling s, grandparent g, out and out's
    // parent -- those it must own exclusively (atomicMax of (gain, id): the best move wins, deterministic).  The nodes BETWEEN them (up to
    // the pivot and down to out) only have their boxes changed, which the refit redoes anyway; what they must not be is MOVED by somebody
    // else (two moves that carry each other's target subtree away would close a cycle), so a move also fails when any node on its path
    // is claimed by another move.  Many moves may pass THROUGH the same upper nodes, unlike with whole-path locks (lockmode=path).
    std::vector<std::atomic<uint64_t>> lock(N);
    for (auto& l : lock)
        l.store(0, std::memory_order_relaxed);
    auto key_of = [&](int x) {
        uint32_t gb;
        memcpy(&gb, &mv[x].gain, 4);
        return ((uint64_t)gb << 32) | (uint32_t)x;
    };
    auto for_topo = [&](int x, auto&& fn) {
        const int p = t.parent[x];
        fn(x), fn(p), fn(sibling(t, x)), fn(t.parent[p]), fn(mv[x].out), fn(t.parent[mv[x].out]);
    };
    auto for_path = [&](int x, auto&& fn) {
        const Move& m = mv[x];
        for (int a = t.parent[x]; a != m.pivot; a = t.parent[a])
            fn(a);
        fn(m.pivot);
        for (int a = m.out; a != m.pivot; a = t.parent[a])
            fn(a);
    };
    long cand = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : cand)
    for (int x = 0; x < N; ++x)
        if (mv[x].out >= 0)
        {
            ++cand;
            const uint64_t k = key_of(x);
            auto claim = [&](int a) {
                uint64_t cur = lock[a].load(std::memory_order_relaxed);
                while (cur < k && !lock[a].compare_exchange_weak(cur, k, std::memory_order_relaxed))
                {
                }
            };
            for_topo(x, claim);
            if (pathLocks)
                for_path(x, claim);
        }
    std::vector<int> winners;
    if (pathLocks)
    {
        for (int x = 0; x < N; ++x)
            if (mv[x].out >= 0)
            {
                const uint64_t k = key_of(x);
                bool ok = true;
                for_topo(x, [&](int a) { ok = ok && lock[a].load(std::memory_order_relaxed) == k; });
                for_path(x, [&](int a) { ok = ok && lock[a].load(std::memory_order_relaxed) == k; });
                if (ok)
                    winners.push_back(x);
            }
    }
    else
    {
        // semi-winners own their six nodes; they announce the subtree they carry away (moving[in] = key); a semi-winner whose target lies
        // inside a subtree that a BETTER move carries away (a moving node among the ancestors of `out` below the pivot) steps back
        std::vector<uint64_t> moving(N, 0);
        std::vector<char> semi(N, 0);
#pragma omp parallel for schedule(dynamic, 1024)
        for (int x = 0; x < N; ++x)
            if (mv[x].out >= 0)
            {
                const uint64_t k = key_of(x);
                bool ok = true;
                for_topo(x, [&](int a) { ok = ok && lock[a].load(std::memory_order_relaxed) == k; });
                if (ok)
                    semi[x] = 1, moving[x] = k;
            }
        for (int x = 0; x < N; ++x)
            if (semi[x])
            {
                const uint64_t k = key_of(x);
                bool ok = true;
                for (int a = t.parent[mv[x].out]; a != mv[x].pivot; a = t.parent[a])
                    ok = ok && !(moving[a] > k);
                if (ok)
                    winners.push_back(x);
            }
    }
    if (candOut)
        *candOut = cand;
    double gs = 0;
    for (int x : winners)
    {
        const int out = mv[x].out;
        const int p = t.parent[x], s = sibling(t, x), g = t.parent[p];
        // take p (with x) out: s replaces p under g
        (t.L[g] == p ? t.L[g] : t.R[g]) = s;
        t.parent[s] = g;
        // p goes in above `out`
        const int po = t.parent[out];
        (t.L[po] == out ? t.L[po] : t.R[po]) = p;
        t.parent[p] = po;
        t.L[p] = x, t.R[p] = out;
        t.parent[out] = p, t.parent[x] = p;
        gs += mv[x].gain;
    }
    if (g_sparse)
    {
        std::vector<char> next(N, 0);
        for (int x = 0; x < N; ++x)
            if (mv[x].out >= 0)
                next[x] = 1; // wanted to move (won or lost): look again
        for (int x : winners)
        {
            // the nodes along the old and the new place's paths to the root got new boxes: they and their children may want to move now
            for (int a = t.parent[x]; a >= 0; a = t.parent[a])
            {
                if (next[a] == 2)
                    break;
                next[a] = 2;
                next[t.L[a]] = std::max<char>(next[t.L[a]], 1), next[t.R[a]] = std::max<char>(next[t.R[a]], 1);
            }
        }
        long na = 0;
        for (int x = 0; x < N; ++x)
            na += next[x] ? 1 : 0;
        printf("  next round searches %ld of %d nodes\n", na, N);
        g_active.swap(next);
    }
    refit(t);
    if (gainSum)
        *gainSum = gs;
    if (visitsOut)
        *visitsOut = visits;
    return (int)winners.size();
}

// ---- 4-wide collapse (greedy by area = k_collapse) and the flattened structure the simulator walks ----
struct Wide
{
    Box cb[4];
    int ref[4]; // >= 0 wide node, < 0: ~((first << 3) | (count - 1)), INT_MIN: empty
    int cnt;
};
struct Flat
{
    std::vector<Wide> nodes;
    std::vector<uint32_t> leafPrim;
};
static int quantMode = 0;
static void collapse(const Tree& t, int leafMax, Flat& f, bool sahRule)
{
    const int n = t.n;
    f.nodes.clear();
    f.leafPrim.clear();
    f.nodes.reserve(n);
    f.leafPrim.reserve(n);
    auto sz = [&](int c) { return c >= n - 1 ? 1 : t.size[c]; };
    auto openable = [&](int c) { return c < n - 1 && t.size[c] > leafMax; };
    struct Item
    {
        int bin, out;
    };
    std::vector<Item> q{ Item{ t.root, 0 } };
    f.nodes.push_back(Wide());
    for (size_t qi = 0; qi < q.size(); ++qi)
    {
        const Item it = q[qi];
        int slot[4], cnt = 2;
        slot[0] = t.L[it.bin], slot[1] = t.R[it.bin];
        while (cnt < 4)
        {
            int best = -1;
            float bestA = -1.0f;
            for (int k = 0; k < cnt; ++k)
                if (openable(slot[k]))
                {
                    float a = area(t.box[slot[k]]);
                    if (sahRule) // open the child whose opening removes the most area: A(c) - is always paid; prefer children whose own children are small
                        a = a - 0.5f * (area(t.box[t.L[slot[k]]]) + area(t.box[t.R[slot[k]]]));
                    if (a > bestA)
                        bestA = a, best = k;
                }
            if (best < 0)
                break;
            const int c = slot[best];
            for (int k = cnt; k > best + 1; --k)
                slot[k] = slot[k - 1];
            slot[best] = t.L[c], slot[best + 1] = t.R[c];
            ++cnt;
        }
        Wide w;
        w.cnt = cnt;
        for (int k = 0; k < 4; ++k)
            w.ref[k] = INT32_MIN, w.cb[k] = empty_box();
        for (int k = 0; k < cnt; ++k)
        {
            const int c = slot[k];
            w.cb[k] = t.box[c];
            if (!openable(c))
            {
                const int first = (int)f.leafPrim.size();
                std::vector<int> st{ c };
                while (!st.empty())
                {
                    const int x = st.back();
                    st.pop_back();
                    if (x >= n - 1)
                        f.leafPrim.push_back(t.prim[x - (n - 1)]);
                    else
                        st.push_back(t.R[x]), st.push_back(t.L[x]);
                }
                w.ref[k] = ~((first << 3) | (sz(c) - 1));
            }
            else
            {
                w.ref[k] = (int)f.nodes.size();
                f.nodes.push_back(Wide());
                q.push_back(Item{ c, w.ref[k] });
            }
        }
        if (quantMode)
        {
            // the GPU node's 8-bit child boxes: origin = node box min, one cell size per axis covering the extent in 255 steps -- a power of two
            // (quantMode 1, what encode_node4 stores) or the extent / 255 itself (quantMode 2)
            const Box nb = t.box[it.bin];
            for (int a = 0; a < 3; ++a)
            {
                const float ext = std::max(nb.hi[a] - nb.lo[a], 1e-30f);
                float cell = ext / 255.0f;
                if (quantMode == 1)
                {
                    int e;
                    (void)std::frexp(cell, &e);
                    cell = std::ldexp(1.0f, e); // 255 * 2^e > ext
                }
                else
                    cell *= 1.000001f;
                for (int k = 0; k < cnt; ++k)
                {
                    const float ql = std::floor((w.cb[k].lo[a] - nb.lo[a]) / cell), qh = std::ceil((w.cb[k].hi[a] - nb.lo[a]) / cell);
                    w.cb[k].lo[a] = nb.lo[a] + std::min(std::max(ql, 0.0f), 255.0f) * cell;
                    w.cb[k].hi[a] = nb.lo[a] + std::min(std::max(qh, 0.0f), 255.0f) * cell;
                }
            }
        }
        f.nodes[it.out] = w;
    }
}

// ---- cost-optimal collapse (after Ylitie, Karras & Laine 2017, section 4.1, for 4-wide nodes) ----
// C(x, i) = cheapest way to represent x's subtree by at most i wide-node children (i = 1 .. 3): a leaf (<= leafMax primitives), one wide node whose
// four slots are dealt between x's two children, or -- for i > 1 -- x dissolved and the i slots dealt between its children.  cn / ct price a wide-node visit
// and a primitive test per unit of box area.  The top-down pass follows the argmins.
static void collapse_dp(const Tree& t, int leafMax, Flat& f, float cn, float ct)
{
    const int n = t.n, total = 2 * n - 1;
    std::vector<std::array<float, 3>> C(total);
    std::vector<int> order; // internal nodes, parents before children
    order.reserve(n);
    order.push_back(t.root);
    for (size_t i = 0; i < order.size(); ++i)
    {
        const int x = order[i];
        if (t.L[x] < n - 1)
            order.push_back(t.L[x]);
        if (t.R[x] < n - 1)
            order.push_back(t.R[x]);
    }
    for (int x = n - 1; x < total; ++x)
        C[x] = { area(t.box[x]) * ct, area(t.box[x]) * ct, area(t.box[x]) * ct };
    auto deal = [&](int x, int j) { // j slots between the children of x
        float best = 3e38f;
        for (int k = 1; k < j; ++k)
            best = std::min(best, C[t.L[x]][k - 1] + C[t.R[x]][j - k - 1]);
        return best;
    };
    for (size_t i = order.size(); i-- > 0;)
    {
        const int x = order[i];
        const float A = area(t.box[x]);
        const float leaf = t.size[x] <= leafMax ? A * t.size[x] * ct : 3e38f;
        const float inner = A * cn + deal(x, 4);
        C[x][0] = std::min(leaf, inner);
        C[x][1] = std::min(C[x][0], deal(x, 2));
        C[x][2] = std::min(C[x][1], deal(x, 3));
    }
    f.nodes.clear();
    f.leafPrim.clear();
    struct Item
    {
        int bin, out;
    };
    std::vector<Item> q{ Item{ t.root, 0 } };
    f.nodes.push_back(Wide());
    long nLeaf = 0, nLeafPrims = 0, slotsUsed = 0;
    for (size_t qi = 0; qi < q.size(); ++qi)
    {
        const Item it = q[qi];
        int slot[4], cnt = 0;
        // expand (x, budget) pairs into the wide node's children
        struct E
        {
            int x, b;
        };
        std::vector<E> st;
        auto push_deal = [&](int x, int j) {
            int bk = 1;
            float best = 3e38f;
            for (int k = 1; k < j; ++k)
            {
                const float c = C[t.L[x]][k - 1] + C[t.R[x]][j - k - 1];
                if (c < best)
                    best = c, bk = k;
            }
            st.push_back(E{ t.R[x], j - bk }), st.push_back(E{ t.L[x], bk });
        };
        push_deal(it.bin, 4);
        while (!st.empty())
        {
            E e = st.back();
            st.pop_back();
            while (e.b > 1 && (e.x >= n - 1 || C[e.x][e.b - 1] == C[e.x][e.b - 2]))
                --e.b;
            if (e.b == 1)
                slot[cnt++] = e.x;
            else
                push_deal(e.x, e.b);
        }
        Wide w;
        w.cnt = cnt;
        slotsUsed += cnt;
        for (int k = 0; k < 4; ++k)
            w.ref[k] = INT32_MIN, w.cb[k] = empty_box();
        for (int k = 0; k < cnt; ++k)
        {
            const int c = slot[k];
            w.cb[k] = t.box[c];
            const int csz = c >= n - 1 ? 1 : t.size[c];
            const bool isLeaf = c >= n - 1 || (csz <= leafMax && area(t.box[c]) * csz * ct <= area(t.box[c]) * cn + deal(c, 4));
            if (isLeaf)
            {
                const int first = (int)f.leafPrim.size();
                std::vector<int> s2{ c };
                while (!s2.empty())
                {
                    const int x = s2.back();
                    s2.pop_back();
                    if (x >= n - 1)
                        f.leafPrim.push_back(t.prim[x - (n - 1)]);
                    else
                        s2.push_back(t.R[x]), s2.push_back(t.L[x]);
                }
                w.ref[k] = ~((first << 3) | (csz - 1));
                ++nLeaf, nLeafPrims += csz;
            }
            else
            {
                w.ref[k] = (int)f.nodes.size();
                f.nodes.push_back(Wide());
                q.push_back(Item{ c, w.ref[k] });
            }
        }
        f.nodes[it.out] = w;
    }
    printf("dp collapse: cost %.4g, %.2f children per node, %.2f primitives per leaf\n", (double)C[t.root][0] / area(t.box[t.root]), (double)slotsUsed / f.nodes.size(),
           (double)nLeafPrims / nLeaf);
}

// ---- traversal simulator ----
struct Ray
{
    float o[3], d[3], tmax;
};
struct Tri
{
    float v[9];
};
static inline bool tri_hit(const Tri& tr, const Ray& r, float tmaxv, float& t)
{
    const float* a = tr.v;
    const float e1[3] = { a[3] - a[0], a[4] - a[1], a[5] - a[2] }, e2[3] = { a[6] - a[0], a[7] - a[1], a[8] - a[2] };
    const float p[3] = { r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0] };
    const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (std::fabs(det) < 1e-20f)
        return false;
    const float inv = 1.0f / det;
    const float s[3] = { r.o[0] - a[0], r.o[1] - a[1], r.o[2] - a[2] };
    const float u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inv;
    if (u < 0 || u > 1)
        return false;
    const float q[3] = { s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0] };
    const float v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) * inv;
    if (v < 0 || u + v > 1)
        return false;
    t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    return t > 1e-4f && t < tmaxv;
}
struct Counts
{
    double nodes = 0, tris = 0;
    long rays = 0, hits = 0;
};
// closest-hit (children sorted by entry distance) or any-hit (stored order, first hit ends the ray)
// (study, `hist=1`: node visits by the number of children the ray enters -- 0 = the visit was a dead end -- and whether the ray starts INSIDE the node's box)
static long g_hitHist[2][5];
static bool g_hist = false;
static bool trace(const Flat& f, const std::vector<Tri>& tris, const Ray& r, bool anyHit, float& tHit, uint32_t& primHit, long& nNodes, long& nTris)
{
    float inv[3];
    for (int k = 0; k < 3; ++k)
        inv[k] = 1.0f / (std::fabs(r.d[k]) > 1e-30f ? r.d[k] : (r.d[k] < 0 ? -1e-30f : 1e-30f));
    int stack[256], sp = 0, cur = 0;
    float best = r.tmax;
    bool found = false;
    for (;;)
    {
        if (cur >= 0)
        {
            const Wide& w = f.nodes[cur];
            ++nNodes;
            float tn[4];
            int rf[4];
            int c = 0;
            for (int k = 0; k < w.cnt; ++k)
            {
                float t0 = 0.0f, t1 = best;
                for (int a = 0; a < 3; ++a)
                {
                    const float x0 = (w.cb[k].lo[a] - r.o[a]) * inv[a], x1 = (w.cb[k].hi[a] - r.o[a]) * inv[a];
                    t0 = std::max(t0, std::min(x0, x1)), t1 = std::min(t1, std::max(x0, x1));
                }
                if (t0 <= t1)
                    tn[c] = t0, rf[c] = w.ref[k], ++c;
            }
            if (g_hist)
            {
                bool inside = true; // the union of the children's boxes around the origin?
                Box u = empty_box();
                for (int k = 0; k < w.cnt; ++k)
                    u = merge(u, w.cb[k]);
                for (int a = 0; a < 3; ++a)
                    inside = inside && r.o[a] >= u.lo[a] && r.o[a] <= u.hi[a];
#pragma omp atomic
                g_hitHist[inside ? 1 : 0][c]++;
            }
            if (!anyHit)
                for (int i = 1; i < c; ++i) // insertion sort, nearest first
                    for (int j = i; j > 0 && tn[j] < tn[j - 1]; --j)
                        std::swap(tn[j], tn[j - 1]), std::swap(rf[j], rf[j - 1]);
            for (int i = c - 1; i >= 1; --i)
                stack[sp++] = rf[i];
            if (c)
            {
                cur = rf[0];
                continue;
            }
        }
        else
        {
            const uint32_t e = (uint32_t)~cur;
            const uint32_t first = e >> 3, count = (e & 7u) + 1u;
            for (uint32_t k = 0; k < count; ++k)
            {
                ++nTris;
                float t;
                if (tri_hit(tris[f.leafPrim[first + k]], r, best, t))
                    best = t, found = true, primHit = f.leafPrim[first + k];
            }
            if (anyHit && found)
                break;
        }
        if (!sp)
            break;
        cur = stack[--sp];
    }
    tHit = best;
    return found;
}

static uint32_t rng_state(uint32_t& s)
{
    s ^= s << 13, s ^= s >> 17, s ^= s << 5;
    return s;
}
static float rnd(uint32_t& s)
{
    return (rng_state(s) >> 8) * (1.0f / 16777216.0f);
}

int main(int argc, char** argv)
{
    if (argc < 2)
        return 1;
    std::map<std::string, std::string> opt;
    for (int i = 2; i < argc; ++i)
    {
        std::string a = argv[i];
        const size_t e = a.find('=');
        opt[a.substr(0, e)] = e == std::string::npos ? "1" : a.substr(e + 1);
    }
    auto geti = [&](const char* k, int d) { return opt.count(k) ? atoi(opt[k].c_str()) : d; };
    auto getf = [&](const char* k, float d) { return opt.count(k) ? (float)atof(opt[k].c_str()) : d; };
    const std::string base = argv[1];
    std::vector<Tri> tris;
    {
        FILE* fp = fopen((base + ".tris").c_str(), "rb");
        fseek(fp, 0, SEEK_END);
        const long sz = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        tris.resize(sz / sizeof(Tri));
        if (fread(tris.data(), sizeof(Tri), tris.size(), fp) != tris.size())
            return 2;
        fclose(fp);
    }
    std::vector<Ray> cam;
    {
        FILE* fp = fopen((base + ".rays").c_str(), "rb");
        fseek(fp, 0, SEEK_END);
        const long sz = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        std::vector<float> raw(sz / 4);
        if (fread(raw.data(), 4, raw.size(), fp) != raw.size())
            return 2;
        fclose(fp);
        const int nr = std::min<int>((int)raw.size() / 6, geti("rays", 100000));
        for (int i = 0; i < nr; ++i)
            cam.push_back(Ray{ { raw[6 * i], raw[6 * i + 1], raw[6 * i + 2] }, { raw[6 * i + 3], raw[6 * i + 4], raw[6 * i + 5] }, 1e16f });
    }
    const int nT = (int)tris.size();
    printf("%d triangles, %zu camera rays, %d threads\n", nT, cam.size(), omp_get_max_threads());

    // ---- references (pre-splitting: a triangle whose box is large against its area is cut into several boxes) ----
    std::vector<Box> pb;
    std::vector<uint32_t> primOf;
    {
        const float budget = getf("split", 0.0f); // extra references as a fraction of the triangles
        auto tri_box = [&](const Tri& t) {
            Box b = empty_box();
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k)
                    b.lo[k] = std::min(b.lo[k], t.v[3 * v + k]), b.hi[k] = std::max(b.hi[k], t.v[3 * v + k]);
            return b;
        };
        if (budget <= 0.0f)
        {
            pb.resize(nT), primOf.resize(nT);
            for (int i = 0; i < nT; ++i)
                pb[i] = tri_box(tris[i]), primOf[i] = (uint32_t)i;
        }
        else
        {
            // priority of a triangle = box half-area - 2 * triangle area (the empty part of its box, what a split can remove); the budget is
            // handed out in proportion: splits_i = floor(priority_i / total * budget * n), each split halves the polygon's box along its longest
            // axis at the midpoint (Ganestam & Doggett 2016 style recursive clipping)
            std::vector<float> pr(nT);
            const float splitExp = getf("splitexp", 1.0f / 3.0f);
            double tot = 0;
            for (int i = 0; i < nT; ++i)
            {
                const Tri& t = tris[i];
                const float e1[3] = { t.v[3] - t.v[0], t.v[4] - t.v[1], t.v[5] - t.v[2] }, e2[3] = { t.v[6] - t.v[0], t.v[7] - t.v[1], t.v[8] - t.v[2] };
                const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                // projected areas: the box half-area equals sum of |projections| only for a box-filling shape; priority as in the literature:
                const float triA = std::fabs(cx) + std::fabs(cy) + std::fabs(cz); // 2 x the sum of the three axis projections of the triangle
                pr[i] = std::pow(std::max(0.0f, area(tri_box(t)) - 0.5f * triA), splitExp); // X^(1/3) as in Fuetterling et al. / embree presplit heuristics
                tot += pr[i];
            }
            // scale such that the extra references come out at the budget (bisection on the floor sum)
            double slo = 0, shi = budget * nT / tot * 64;
            for (int it = 0; it < 40; ++it)
            {
                const double mid = 0.5 * (slo + shi);
                double extra = 0;
                for (int i = 0; i < nT; ++i)
                    extra += std::floor(pr[i] * mid);
                (extra > budget * nT ? shi : slo) = mid;
            }
            const double scale = slo;
            for (int i = 0; i < nT; ++i)
            {
                const int pieces = 1 + (int)(pr[i] * scale);
                // recursive midpoint clipping of the triangle polygon into `pieces` boxes
                struct Poly
                {
                    std::vector<std::array<float, 3>> v;
                    int budget;
                };
                std::vector<Poly> work;
                Poly p0;
                for (int v = 0; v < 3; ++v)
                    p0.v.push_back({ tris[i].v[3 * v], tris[i].v[3 * v + 1], tris[i].v[3 * v + 2] });
                p0.budget = pieces;
                work.push_back(p0);
                while (!work.empty())
                {
                    Poly p = work.back();
                    work.pop_back();
                    Box b = empty_box();
                    for (auto& v : p.v)
                        for (int k = 0; k < 3; ++k)
                            b.lo[k] = std::min(b.lo[k], v[k]), b.hi[k] = std::max(b.hi[k], v[k]);
                    if (p.budget <= 1 || p.v.size() < 3)
                    {
                        pb.push_back(b), primOf.push_back((uint32_t)i);
                        continue;
                    }
                    int ax = 0;
                    for (int k = 1; k < 3; ++k)
                        if (b.hi[k] - b.lo[k] > b.hi[ax] - b.lo[ax])
                            ax = k;
                    const float mid = 0.5f * (b.lo[ax] + b.hi[ax]);
                    Poly lo, hi;
                    const size_t nv = p.v.size();
                    for (size_t a = 0; a < nv; ++a)
                    {
                        const auto& A = p.v[a];
                        const auto& Bv = p.v[(a + 1) % nv];
                        if (A[ax] <= mid)
                            lo.v.push_back(A);
                        if (A[ax] >= mid)
                            hi.v.push_back(A);
                        if ((A[ax] < mid && Bv[ax] > mid) || (A[ax] > mid && Bv[ax] < mid))
                        {
                            const float f = (mid - A[ax]) / (Bv[ax] - A[ax]);
                            std::array<float, 3> X = { A[0] + f * (Bv[0] - A[0]), A[1] + f * (Bv[1] - A[1]), A[2] + f * (Bv[2] - A[2]) };
                            X[ax] = mid;
                            lo.v.push_back(X), hi.v.push_back(X);
                        }
                    }
                    lo.budget = p.budget / 2, hi.budget = p.budget - lo.budget;
                    work.push_back(lo), work.push_back(hi);
                }
            }
            printf("pre-split: %zu references (+%.1f %%)\n", pb.size(), 100.0 * (pb.size() - nT) / nT);
        }
    }
    Tree t;
    double t0 = now();
    build_ploc(t, pb, primOf, geti("radius", 12));
    refit(t);
    printf("PLOC: %.2f s, SAH(internal area / root area) = %.2f\n", now() - t0, sah_internal(t));

    ancestors = geti("ancestors", 0) != 0;
    g_sparse = geti("sparse", 0) != 0;
    const int rounds = geti("reinsert", 0), minSize = geti("minsize", 1), stride = geti("stride", 1);
    for (int k = 0; k < rounds; ++k)
    {
        double gs = 0;
        long visits = 0;
        t0 = now();
        long cand = 0;
        if (geti("fullevery", 0) > 0 && (k % geti("fullevery", 0)) == 0)
            g_active.clear(); // a full search again
        const int moved = reinsertion_batch(t, minSize, stride, k % stride, &gs, &visits, opt.count("lockmode") && opt["lockmode"] == "path", &cand);
        printf("reinsertion %d: %ld candidates, %d moved, %.1f M search visits, %.2f s, SAH = %.2f\n", k, cand, moved, visits * 1e-6, now() - t0, sah_internal(t));
    }

    g_hist = geti("hist", 0) != 0;
    quantMode = geti("quant", 0);
    Flat f;
    if (opt.count("collapse") && opt["collapse"] == "dp")
        collapse_dp(t, geti("leaf", 2), f, getf("cn", 1.0f), getf("ct", 1.0f));
    else
        collapse(t, geti("leaf", 2), f, opt.count("collapse") && opt["collapse"] == "sah");
    printf("4-wide: %zu nodes, %zu leaf slots\n", f.nodes.size(), f.leafPrim.size());

    // ---- rays: camera, one diffuse bounce from the camera hits, shadow rays from those hits towards a point under the ceiling ----
    Box sb = t.box[t.root];
    const float lightP[3] = { 0.5f * (sb.lo[0] + sb.hi[0]), sb.lo[1] + 0.9f * (sb.hi[1] - sb.lo[1]), 0.5f * (sb.lo[2] + sb.hi[2]) };
    Counts cc, cb2, cs;
    const int nr = (int)cam.size();
    std::vector<Ray> bounce(nr), shadow(nr);
    std::vector<char> ok(nr, 0);
    double cn = 0, ct = 0, bn = 0, bt = 0, sn = 0, stt = 0;
    long hits = 0, bhits = 0, socc = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : cn, ct, hits)
    for (int i = 0; i < nr; ++i)
    {
        float th;
        uint32_t ph = 0;
        long a = 0, b = 0;
        if (trace(f, tris, cam[i], false, th, ph, a, b))
        {
            ++hits;
            const Tri& tr = tris[ph];
            const float e1[3] = { tr.v[3] - tr.v[0], tr.v[4] - tr.v[1], tr.v[5] - tr.v[2] }, e2[3] = { tr.v[6] - tr.v[0], tr.v[7] - tr.v[1], tr.v[8] - tr.v[2] };
            float nrm[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
            const float l = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]) + 1e-30f;
            const float dn = nrm[0] * cam[i].d[0] + nrm[1] * cam[i].d[1] + nrm[2] * cam[i].d[2];
            for (int k = 0; k < 3; ++k)
                nrm[k] = (dn > 0 ? -nrm[k] : nrm[k]) / l;
            uint32_t s = 0x9e3779b9u * (uint32_t)(i + 1) | 1u;
            // cosine-distributed direction about the normal
            const float u1 = rnd(s), u2 = rnd(s);
            const float rr = std::sqrt(u1), ph2 = 6.2831853f * u2;
            float tx[3] = { std::fabs(nrm[0]) < 0.9f ? 1.0f : 0.0f, std::fabs(nrm[0]) < 0.9f ? 0.0f : 1.0f, 0.0f };
            float bx[3] = { tx[1] * nrm[2] - tx[2] * nrm[1], tx[2] * nrm[0] - tx[0] * nrm[2], tx[0] * nrm[1] - tx[1] * nrm[0] };
            const float bl = std::sqrt(bx[0] * bx[0] + bx[1] * bx[1] + bx[2] * bx[2]);
            for (int k = 0; k < 3; ++k)
                bx[k] /= bl;
            const float by[3] = { nrm[1] * bx[2] - nrm[2] * bx[1], nrm[2] * bx[0] - nrm[0] * bx[2], nrm[0] * bx[1] - nrm[1] * bx[0] };
            Ray br, sr;
            float dl = 0;
            for (int k = 0; k < 3; ++k)
            {
                br.o[k] = cam[i].o[k] + th * cam[i].d[k] + 1e-3f * nrm[k];
                br.d[k] = rr * std::cos(ph2) * bx[k] + rr * std::sin(ph2) * by[k] + std::sqrt(std::max(0.0f, 1.0f - u1)) * nrm[k];
                sr.o[k] = br.o[k];
                sr.d[k] = lightP[k] - br.o[k];
                dl += sr.d[k] * sr.d[k];
            }
            dl = std::sqrt(dl);
            for (int k = 0; k < 3; ++k)
                sr.d[k] /= dl;
            br.tmax = 1e16f, sr.tmax = dl;
            bounce[i] = br, shadow[i] = sr, ok[i] = 1;
        }
        cn += a, ct += b;
    }
    long nb = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : bn, bt, sn, stt, bhits, socc, nb)
    for (int i = 0; i < nr; ++i)
        if (ok[i])
        {
            float th;
            uint32_t ph = 0;
            long a = 0, b = 0;
            bhits += trace(f, tris, bounce[i], false, th, ph, a, b) ? 1 : 0;
            bn += a, bt += b;
            a = b = 0;
            socc += trace(f, tris, shadow[i], true, th, ph, a, b) ? 1 : 0;
            sn += a, stt += b;
            ++nb;
        }
    if (g_hist)
        for (int in = 0; in < 2; ++in)
        {
            long tot = 0;
            for (int k = 0; k < 5; ++k)
                tot += g_hitHist[0][k] + g_hitHist[1][k];
            printf("node visits with the origin %s the node: children entered 0 / 1 / 2 / 3 / 4 = %.1f / %.1f / %.1f / %.1f / %.1f %% of all visits\n", in ? "INSIDE " : "outside",
                   100.0 * g_hitHist[in][0] / tot, 100.0 * g_hitHist[in][1] / tot, 100.0 * g_hitHist[in][2] / tot, 100.0 * g_hitHist[in][3] / tot, 100.0 * g_hitHist[in][4] / tot);
        }
    printf("camera  %8d rays: %6.2f nodes %5.2f tris per ray (%.1f %% hit)\n", nr, cn / nr, ct / nr, 100.0 * hits / nr);
    printf("bounce  %8ld rays: %6.2f nodes %5.2f tris per ray (%.1f %% hit)\n", nb, bn / nb, bt / nb, 100.0 * bhits / nb);
    printf("shadow  %8ld rays: %6.2f nodes %5.2f tris per ray (%.1f %% occluded)\n", nb, sn / nb, stt / nb, 100.0 * socc / nb);
    return 0;
}
