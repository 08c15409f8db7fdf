// A second, device-private BVH over the same triangles ("entry BVH"), built with a binned surface-area heuristic.
//
// Why: the reference's buildBVHwithSAH falls back to an x-median split whenever every candidate cost reaches its
// INF = 114514 (c5:500-502) — on large scenes that is every node near the root, and a ray that escapes must then cross
// ~1000 slab boxes.  Results must stay the reference's, so the reference tree is still the one that decides every hit
// (rt_device.hip walks it with a distance bound); this tree only supplies that bound, or proves there is nothing to hit.
// It is never visible at the ABI and has no counterpart in the reference.  Nothing here needs bit-exactness: boxes are
// plain min / max over vertices (so every box contains its triangles and its children's boxes exactly, which is all the
// conservative box test of rt_trav.h needs), the split choice only affects speed.  The result is deterministic all the
// same (it does not depend on the number of threads): chunk results are merged in chunk order, the parallel partition is
// a stable one.
//
// Every start of a program with a large scene pays this build (the on-disk cache holds the reference's arrays only), so
// it works on 32-byte (box, triangle) items that are partitioned in place — every pass streams memory — and ranges of at
// least kParMin items are bounded, binned and partitioned by all threads (chunks as OpenMP tasks); smaller ranges are one
// task each.  10 M triangles: 0.7 s on 16 cores.
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
constexpr int kParMin = 1 << 17;  // ranges at least this long are processed by all threads, in chunks of kChunk
constexpr int kChunk = 1 << 15;

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

struct Item {  // one triangle: its box and its index in the caller's array
    float lo[3], hi[3];
    int tri, pad;
};
static_assert(sizeof(Item) == 32, "items are streamed and swapped as 32-byte records");

struct Ctx {
    Item* items;           // permuted in place
    Item* tmp;             // scratch for the parallel stable partition (as long as items)
    FastBvhNode* nodes;    // preallocated 2 * n
    std::atomic<int> next{0};
};

template <class F>
void par_chunks(int n_chunks, F&& f) {
#pragma omp taskgroup
    for (int c = 0; c < n_chunks; ++c) {
#pragma omp task default(shared) firstprivate(c)
        f(c);
    }
}

inline void centroid(const Item& t, float ctr[3]) {
    ctr[0] = 0.5f * (t.lo[0] + t.hi[0]), ctr[1] = 0.5f * (t.lo[1] + t.hi[1]), ctr[2] = 0.5f * (t.lo[2] + t.hi[2]);
}

struct Binning {
    int axis;
    float k1, c0;
    int bin_of(const Item& t) const {
        const int k = (int)(k1 * (0.5f * (t.lo[axis] + t.hi[axis]) - c0));
        return k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
    }
};

struct Sub {  // what a finished subtree reports upwards
    int id, n_internal, depth;
    double area;  // sum of the areas of its internal nodes' boxes
};

Sub build(Ctx& c, int l, int r, int depth) {  // [l, r)
    const int id = c.next.fetch_add(1);
    FastBvhNode& nd = c.nodes[id];
    const int n = r - l;
    const bool big = n >= kParMin;
    const int nch = big ? (n + kChunk - 1) / kChunk : 1;
    Box b, cb;
    b.reset(), cb.reset();
    auto bounds = [&](int lo, int hi, Box& bx, Box& cbx) {
        for (int i = lo; i < hi; ++i) {
            const Item& t = c.items[i];
            bx.grow(t.lo, t.hi);
            float ctr[3];
            centroid(t, ctr);
            cbx.grow(ctr, ctr);
        }
    };
    if (big) {
        std::vector<Box> pb(nch), pcb(nch);
        par_chunks(nch, [&](int k) {
            pb[k].reset(), pcb[k].reset();
            bounds(l + k * kChunk, std::min(l + (k + 1) * kChunk, r), pb[k], pcb[k]);
        });
        for (int k = 0; k < nch; ++k) b.grow(pb[k]), cb.grow(pcb[k]);
    } else {
        bounds(l, r, b, cb);
    }
    memcpy(nd.AA, b.lo, 12), memcpy(nd.BB, b.hi, 12);
    nd.left = nd.right = -1, nd.first = l, nd.count = 0;
    if (n <= kLeafMax) {
        nd.count = n;
        return Sub{id, 0, 0, 0.0};
    }
    // binned SAH on the axis with the widest centroid spread; falls back to a median split of the index range when the
    // centroids coincide or no split beats the leaf
    int axis = 0;
    float ext = cb.hi[0] - cb.lo[0];
    for (int a = 1; a < 3; ++a)
        if (cb.hi[a] - cb.lo[a] > ext) ext = cb.hi[a] - cb.lo[a], axis = a;
    int mid = -1;
    if (ext > 0.0f && ext < INFINITY && depth < kMaxSahDepth) {  // (deeper: median splits only, which bound the depth)
        const Binning bn{axis, (float)kBins * (1.0f - 1e-6f) / ext, cb.lo[axis]};
        Box bb[kBins];
        int cnt[kBins];
        for (int k = 0; k < kBins; ++k) bb[k].reset(), cnt[k] = 0;
        auto bin_range = [&](int lo, int hi, Box* bx, int* cn) {
            for (int i = lo; i < hi; ++i) {
                const Item& t = c.items[i];
                const int k = bn.bin_of(t);
                bx[k].grow(t.lo, t.hi), cn[k]++;
            }
        };
        if (big) {
            std::vector<Box> pb((size_t)nch * kBins);
            std::vector<int> pc((size_t)nch * kBins, 0);
            par_chunks(nch, [&](int k) {
                for (int j = 0; j < kBins; ++j) pb[(size_t)k * kBins + j].reset();
                bin_range(l + k * kChunk, std::min(l + (k + 1) * kChunk, r), &pb[(size_t)k * kBins], &pc[(size_t)k * kBins]);
            });
            for (int k = 0; k < nch; ++k)
                for (int j = 0; j < kBins; ++j) bb[j].grow(pb[(size_t)k * kBins + j]), cnt[j] += pc[(size_t)k * kBins + j];
        } else {
            bin_range(l, r, bb, cnt);
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
            if (big) {  // stable partition through the scratch array: chunk k's left / right items go where a sequential pass would put them
                std::vector<int> nl(nch + 1, 0);
                par_chunks(nch, [&](int k) {
                    int m = 0;
                    for (int i = l + k * kChunk, e = std::min(l + (k + 1) * kChunk, r); i < e; ++i) m += bn.bin_of(c.items[i]) <= best;
                    nl[k + 1] = m;
                });
                for (int k = 0; k < nch; ++k) nl[k + 1] += nl[k];
                const int n_left = nl[nch];
                par_chunks(nch, [&](int k) {
                    const int b0 = l + k * kChunk, e = std::min(b0 + kChunk, r);
                    int wl = l + nl[k], wr = l + n_left + (b0 - l - nl[k]);
                    for (int i = b0; i < e; ++i) {
                        if (bn.bin_of(c.items[i]) <= best) c.tmp[wl++] = c.items[i];
                        else c.tmp[wr++] = c.items[i];
                    }
                });
                par_chunks(nch, [&](int k) {
                    const int b0 = l + k * kChunk, e = std::min(b0 + kChunk, r);
                    memcpy(c.items + b0, c.tmp + b0, (size_t)(e - b0) * sizeof(Item));
                });
                mid = l + n_left;
            } else {
                Item* m = std::partition(c.items + l, c.items + r, [&](const Item& t) { return bn.bin_of(t) <= best; });
                mid = (int)(m - c.items);
            }
        }
    }
    if (mid <= l || mid >= r) mid = l + n / 2;
    Sub left, right;
    if (n >= kTaskMin) {
#pragma omp task shared(left, c) firstprivate(l, mid, depth)
        left = build(c, l, mid, depth + 1);
        right = build(c, mid, r, depth + 1);
#pragma omp taskwait
    } else {
        left = build(c, l, mid, depth + 1);
        right = build(c, mid, r, depth + 1);
    }
    FastBvhNode& me = c.nodes[id];
    me.left = left.id, me.right = right.id;
    me.first = 1 + left.n_internal + right.n_internal;  // (count stays 0: an internal node)
    return Sub{id, me.first, 1 + std::max(left.depth, right.depth), b.area() + left.area + right.area};
}

}  // namespace

// tri_box: per triangle lo[3] hi[3]; `use` (may be NULL): triangles to include (the rest can never be hit).
// Returns the root index (-1 when no triangle is included); `order` receives the triangle permutation the leaves index.
// `depth`: the deepest leaf (root = 0); `sum_internal_area`: sum over the internal nodes of their boxes' areas.
int build_fast_bvh(const float* tri_box, const char* use, int n_tri, RawBuf<FastBvhNode>& nodes, int* n_nodes, std::vector<int>& order,
                   int* depth, double* sum_internal_area) {
    order.clear();
    nodes.alloc(0);
    *n_nodes = 0, *depth = 0, *sum_internal_area = 0.0;
    // the included triangles, in index order (serial count per block, then a parallel fill: deterministic)
    const int nblk = (n_tri + kChunk - 1) / kChunk;
    std::vector<int> base(nblk + 1, 0);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nblk; ++k) {
        int m = 0;
        for (int i = k * kChunk, e = std::min(n_tri, (k + 1) * kChunk); i < e; ++i) m += (!use || use[i]) ? 1 : 0;
        base[k + 1] = m;
    }
    for (int k = 0; k < nblk; ++k) base[k + 1] += base[k];
    const int n = base[nblk];
    if (n == 0) return -1;
    RawBuf<Item> items, tmp;
    items.alloc((size_t)n);
    if (n >= kParMin) tmp.alloc((size_t)n);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nblk; ++k) {
        int w = base[k];
        for (int i = k * kChunk, e = std::min(n_tri, (k + 1) * kChunk); i < e; ++i) {
            if (use && !use[i]) continue;
            Item& t = items[(size_t)w++];
            memcpy(t.lo, tri_box + (size_t)i * 6, 12), memcpy(t.hi, tri_box + (size_t)i * 6 + 3, 12);
            t.tri = i, t.pad = 0;
        }
    }
    nodes.alloc(2 * (size_t)n);  // (a binary tree with n or fewer leaves; not zero-filled: only the nodes handed out are ever read)
    Ctx c;
    c.items = items.data(), c.tmp = tmp.data(), c.nodes = nodes.data();
    Sub root_sub{-1, 0, 0, 0.0};
#pragma omp parallel
#pragma omp single
    root_sub = build(c, 0, n, 0);
    const int root = root_sub.id;
    *n_nodes = c.next.load(), *depth = root_sub.depth, *sum_internal_area = root_sub.area;
    order.resize((size_t)n);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) order[(size_t)i] = items[(size_t)i].tri;
    return root;
}

}  // namespace rta
