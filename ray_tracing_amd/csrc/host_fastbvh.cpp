// A second, device-private BVH over the same triangles ("entry BVH"), built with a binned surface-area heuristic.
//
// Why: the reference's buildBVHwithSAH falls back to an x-median split whenever every candidate cost reaches its
// INF = 114514 (c5:500-502) — on large scenes that is every node near the root, and a ray that escapes must then cross
// ~1000 slab boxes.  Results must stay the reference's, so the reference tree is still the one that decides every hit
// (rt_device.hip walks it with a distance bound); this tree only supplies that bound, or proves there is nothing to hit.
// It is never visible at the ABI and has no counterpart in the reference.  Nothing here needs bit-exactness: boxes are
// plain min / max over vertices (so every box contains its triangles and its children's boxes exactly, which is all the
// conservative box test of rt_trav.h needs), the split choice only affects speed.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>

#include "rt_host.h"

namespace rta {
namespace {

constexpr int kBins = 16;
constexpr int kLeafMax = 4;       // triangles per leaf
constexpr int kMaxSahDepth = 64;  // SAH splits can be lopsided; past this depth only median splits (<= 24 more levels)
constexpr int kTaskMin = 16384;   // ranges at least this long build their two subtrees as parallel tasks

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int c = 0; c < 3; ++c) lo[c] = INFINITY, hi[c] = -INFINITY;
    }
    void grow(const float* l, const float* h) {
        for (int c = 0; c < 3; ++c) lo[c] = l[c] < lo[c] ? l[c] : lo[c], hi[c] = h[c] > hi[c] ? h[c] : hi[c];
    }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    double area() const {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return (x < 0 || y < 0 || z < 0) ? 0.0 : 2.0 * (x * y + x * z + y * z);
    }
};

struct Ctx {
    const float* tb;       // per triangle: lo[3], hi[3] (6 floats)
    int* idx;              // permutation being built
    FastBvhNode* nodes;    // preallocated 2 * n
    std::atomic<int> next{0};
};

int build(Ctx& c, int l, int r, int depth) {  // [l, r)
    const int id = c.next.fetch_add(1);
    FastBvhNode& nd = c.nodes[id];
    Box b, cb;
    b.reset(), cb.reset();
    for (int i = l; i < r; ++i) {
        const float* t = c.tb + (size_t)c.idx[i] * 6;
        b.grow(t, t + 3);
        const float ctr[3] = {0.5f * (t[0] + t[3]), 0.5f * (t[1] + t[4]), 0.5f * (t[2] + t[5])};
        cb.grow(ctr, ctr);
    }
    memcpy(nd.AA, b.lo, 12), memcpy(nd.BB, b.hi, 12);
    nd.left = nd.right = -1, nd.first = l, nd.count = 0;
    const int n = r - l;
    if (n <= kLeafMax) {
        nd.count = n;
        return id;
    }
    // binned SAH on the axis with the widest centroid spread; falls back to a median split of the index range when the
    // centroids coincide or no split beats the leaf
    int axis = 0;
    float ext = cb.hi[0] - cb.lo[0];
    for (int a = 1; a < 3; ++a)
        if (cb.hi[a] - cb.lo[a] > ext) ext = cb.hi[a] - cb.lo[a], axis = a;
    int mid = -1;
    if (ext > 0.0f && ext < INFINITY && depth < kMaxSahDepth) {  // (deeper: median splits only, which bound the depth)
        Box bb[kBins];
        int cnt[kBins];
        for (int k = 0; k < kBins; ++k) bb[k].reset(), cnt[k] = 0;
        const float k1 = (float)kBins * (1.0f - 1e-6f) / ext, c0 = cb.lo[axis];
        auto bin_of = [&](int tri) {
            const float* t = c.tb + (size_t)tri * 6;
            int k = (int)(k1 * (0.5f * (t[axis] + t[axis + 3]) - c0));
            return k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
        };
        for (int i = l; i < r; ++i) {
            const int k = bin_of(c.idx[i]);
            const float* t = c.tb + (size_t)c.idx[i] * 6;
            bb[k].grow(t, t + 3), cnt[k]++;
        }
        double right_area[kBins];
        Box acc;
        acc.reset();
        for (int k = kBins - 1; k > 0; --k) acc.grow(bb[k]), right_area[k] = acc.area();
        acc.reset();
        int lc = 0, best = -1;
        double best_cost = INFINITY;
        for (int k = 0; k + 1 < kBins; ++k) {
            acc.grow(bb[k]), lc += cnt[k];
            if (lc == 0 || lc == n) continue;
            const double cost = acc.area() * lc + right_area[k + 1] * (n - lc);
            if (cost < best_cost) best_cost = cost, best = k;
        }
        if (best >= 0) {
            int* m = std::partition(c.idx + l, c.idx + r, [&](int tri) { return bin_of(tri) <= best; });
            mid = (int)(m - c.idx);
        }
    }
    if (mid <= l || mid >= r) mid = l + n / 2;
    int left, right;
    if (n >= kTaskMin) {
#pragma omp task shared(left, c) firstprivate(l, mid, depth)
        left = build(c, l, mid, depth + 1);
        right = build(c, mid, r, depth + 1);
#pragma omp taskwait
    } else {
        left = build(c, l, mid, depth + 1);
        right = build(c, mid, r, depth + 1);
    }
    c.nodes[id].left = left, c.nodes[id].right = right;
    return id;
}

}  // namespace

// tri_box: per triangle lo[3] hi[3]; `use` (may be NULL): triangles to include (the rest can never be hit).
// Returns the root index (-1 when no triangle is included); `order` receives the triangle permutation the leaves index.
int build_fast_bvh(const float* tri_box, const char* use, int n_tri, std::vector<FastBvhNode>& nodes, std::vector<int>& order) {
    order.clear();
    order.reserve(n_tri);
    for (int i = 0; i < n_tri; ++i)
        if (!use || use[i]) order.push_back(i);
    nodes.clear();
    if (order.empty()) return -1;
    nodes.resize(2 * order.size());
    Ctx c;
    c.tb = tri_box, c.idx = order.data(), c.nodes = nodes.data();
    int root = -1;
#pragma omp parallel
#pragma omp single
    root = build(c, 0, (int)order.size(), 0);
    nodes.resize(c.next.load());
    return root;
}

}  // namespace rta
