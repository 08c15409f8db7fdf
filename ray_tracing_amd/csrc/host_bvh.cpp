// Flattened BVH builders of the reference (buildBVHwithSAH c5:462-593, buildBVH c5:399-460), bit-exact.
//
// The reference sorts 144-byte Triangle structs with std::sort four times per SAH node.  The permutation a std::sort
// call produces depends only on the outcomes of its comparisons, so this builder sorts 16-byte (centroid, index)
// proxies with the same comparator on the same ranges in the same sequence and gathers the triangles once at the end.
// Everything else (fp32 operation order, the INF = 114514 clamps, `float split`, strict `<` tie rules, pre-order node
// numbering, left subtree built before right) follows the reference line by line.  Compile with -ffp-contract=off.
//
// Sibling subtrees work on disjoint ranges, so they are built as OpenMP tasks into a pointer tree, which is then numbered
// in the reference's order (a node before its left subtree, the whole left subtree before the right one).  Inside a node the
// four sorts stay in sequence: each starts from the order the previous one left, as in the reference.
//
// The nodes near the root are where a 10 M-triangle build spends its time if each is one thread's work (the root alone: four
// sorts of the whole array and three sweeps).  Ranges of at least kParNodeMin triangles are therefore processed by all
// threads — with results that are still the sequential ones, bit for bit:
//   * the sorts through exact_sort.h (libstdc++'s own introsort steps, run as tasks on disjoint ranges: the permutation
//     std::sort produces, ties included);
//   * the bounding box and the prefix / suffix boxes in chunks: GLM's min(a, b) keeps `a` unless b < a, so a running
//     minimum is the FIRST occurrence of the smallest value (this only matters for the sign of a zero); a chunk's own
//     fold has the same property, and folding the chunk results in order again keeps the first — the chunked value IS the
//     sequential one, and each chunk then continues the sequential recurrence from it;
//   * the cost sweep per chunk with the strict `<` of c5:566, chunk results combined in order with the same strict `<`.
// Needs finite coordinates (a strict weak order for the sorts, no NaN in the folds); otherwise the sequential path runs.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>

#include "exact_sort.h"
#include "rt_host.h"

namespace rta {
namespace {

constexpr float kINF = 114514.0;  // #define INF 114514.0, c5:19

struct Proxy {
    float cx, cy, cz;  // (p1 + p2 + p3) / vec3(3), c5:155-169
    int idx;           // original triangle
};

inline bool cmpx(const Proxy& a, const Proxy& b) { return a.cx < b.cx; }
inline bool cmpy(const Proxy& a, const Proxy& b) { return a.cy < b.cy; }
inline bool cmpz(const Proxy& a, const Proxy& b) { return a.cz < b.cz; }

struct BNode {
    int left, right;
    vec3 AA, BB;
    int n, index;
};

struct TNode {  // node of the tree as it is built; numbered afterwards
    vec3 AA, BB;
    int n, index;
    TNode *left, *right;
};
constexpr int kTaskMin = 8192;  // ranges at least this long build their two subtrees as parallel tasks
constexpr int kParNodeMin = 1 << 17;  // ranges at least this long are sorted and swept by all threads
constexpr int kChunk = 1 << 15;       // chunk of such a range handled by one task

template <class F>
void par_chunks(int n_chunks, F&& f) {
#pragma omp taskgroup
    for (int c = 0; c < n_chunks; ++c) {
#pragma omp task default(shared) firstprivate(c)
        f(c);
    }
}

struct Builder {
    std::vector<Proxy> px;
    std::vector<vec3> tmin, tmax;  // per original triangle: min/max over p1,p2,p3 as min(p1, min(p2, p3))
    std::vector<BNode> nodes;
    int leaf_n;
    bool par_ok = false;  // every coordinate finite: large ranges may take the chunked path

    TNode* new_node(int l, int r) {
        TNode* nd = new TNode();
        nd->left = nd->right = nullptr;
        nd->n = nd->index = 0;
        nd->AA = v3(1145141919, 1145141919, 1145141919);
        nd->BB = v3(-1145141919, -1145141919, -1145141919);
        if (par_ok && r - l + 1 >= kParNodeMin) {
            const int cnt = r - l + 1, nch = (cnt + kChunk - 1) / kChunk;
            std::vector<vec3> cmin(nch), cmax(nch);
            par_chunks(nch, [&](int c) {
                const int b = l + c * kChunk, e = std::min(b + kChunk, r + 1);
                vec3 mn = tmin[px[b].idx], mx = tmax[px[b].idx];
                for (int i = b + 1; i < e; ++i) mn = min3(mn, tmin[px[i].idx]), mx = max3(mx, tmax[px[i].idx]);
                cmin[c] = mn, cmax[c] = mx;
            });
            for (int c = 0; c < nch; ++c) nd->AA = min3(nd->AA, cmin[c]), nd->BB = max3(nd->BB, cmax[c]);
            return nd;
        }
        for (int i = l; i <= r; i++) {
            const vec3 mn = tmin[px[i].idx], mx = tmax[px[i].idx];
            nd->AA.x = fmin_(nd->AA.x, mn.x), nd->AA.y = fmin_(nd->AA.y, mn.y), nd->AA.z = fmin_(nd->AA.z, mn.z);
            nd->BB.x = fmax_(nd->BB.x, mx.x), nd->BB.y = fmax_(nd->BB.y, mx.y), nd->BB.z = fmax_(nd->BB.z, mx.z);
        }
        return nd;
    }

    // pre-order numbering, c5:466-467 / 587-590: vector index = creation order; frees the tree
    int number(TNode* t) {
        if (!t) return 0;
        const int id = (int)nodes.size();
        nodes.push_back(BNode());
        nodes[id].AA = t->AA, nodes[id].BB = t->BB, nodes[id].n = t->n, nodes[id].index = t->index;
        const int left = number(t->left);
        nodes[id].left = left;
        const int right = number(t->right);
        nodes[id].right = right;
        delete t;
        return id;
    }

    void sort_axis(int axis, int l, int r) {
        Proxy* b = px.data() + l;
        Proxy* e = px.data() + r + 1;
        if (par_ok && r - l + 1 >= kParNodeMin) {  // std::sort's permutation, computed by all threads
            if (axis == 0) exact_sort(b, e, cmpx);
            if (axis == 1) exact_sort(b, e, cmpy);
            if (axis == 2) exact_sort(b, e, cmpz);
            return;
        }
        if (axis == 0) std::sort(b, e, cmpx);
        if (axis == 1) std::sort(b, e, cmpy);
        if (axis == 2) std::sort(b, e, cmpz);
    }

    // One axis of the SAH sweep (c5:509-575) over a large range, in chunks; same values as the sequential loops of build_sah.
    void sweep_chunked(int l, int r, vec3* leftMax, vec3* leftMin, vec3* rightMax, vec3* rightMin, float* cost_out, float* split_out) {
        const int cnt = r - l + 1, nch = (cnt + kChunk - 1) / kChunk;
        std::vector<vec3> fmn(nch), fmx(nch), bmn(nch), bmx(nch);  // per chunk: forward and backward folds
        par_chunks(nch, [&](int c) {
            const int b = l + c * kChunk, e = std::min(b + kChunk, r + 1);
            vec3 mn = tmin[px[b].idx], mx = tmax[px[b].idx];
            for (int i = b + 1; i < e; ++i) mn = min3(mn, tmin[px[i].idx]), mx = max3(mx, tmax[px[i].idx]);
            fmn[c] = mn, fmx[c] = mx;
            mn = tmin[px[e - 1].idx], mx = tmax[px[e - 1].idx];
            for (int i = e - 2; i >= b; --i) mn = min3(mn, tmin[px[i].idx]), mx = max3(mx, tmax[px[i].idx]);
            bmn[c] = mn, bmx[c] = mx;
        });
        std::vector<vec3> pmn(nch), pmx(nch), smn(nch), smx(nch);  // the running values just before / after each chunk
        vec3 amn = v3(kINF), amx = v3(-kINF);
        for (int c = 0; c < nch; ++c) pmn[c] = amn, pmx[c] = amx, amn = min3(amn, fmn[c]), amx = max3(amx, fmx[c]);
        amn = v3(kINF), amx = v3(-kINF);
        for (int c = nch - 1; c >= 0; --c) smn[c] = amn, smx[c] = amx, amn = min3(amn, bmn[c]), amx = max3(amx, bmx[c]);
        par_chunks(nch, [&](int c) {
            const int b = l + c * kChunk, e = std::min(b + kChunk, r + 1);
            vec3 mn = pmn[c], mx = pmx[c];
            for (int i = b; i < e; ++i) {
                mx = max3(mx, tmax[px[i].idx]), mn = min3(mn, tmin[px[i].idx]);
                leftMax[i - l] = mx, leftMin[i - l] = mn;
            }
            mn = smn[c], mx = smx[c];
            for (int i = e - 1; i >= b; --i) {
                mx = max3(mx, tmax[px[i].idx]), mn = min3(mn, tmin[px[i].idx]);
                rightMax[i - l] = mx, rightMin[i - l] = mn;
            }
        });
        std::vector<float> ccost(nch), csplit(nch);
        par_chunks(nch, [&](int c) {
            const int b = l + c * kChunk, e = std::min(b + kChunk, r);  // i <= r - 1
            float cost = kINF;
            float split = l;
            for (int i = b; i < e; ++i) {
                float lenx, leny, lenz;
                const vec3 leftAA = leftMin[i - l], leftBB = leftMax[i - l];
                lenx = leftBB.x - leftAA.x, leny = leftBB.y - leftAA.y, lenz = leftBB.z - leftAA.z;
                const float leftS = 2.0 * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
                const float leftcost = leftS * (i - l + 1);
                const vec3 rightAA = rightMin[i - l + 1], rightBB = rightMax[i - l + 1];
                lenx = rightBB.x - rightAA.x, leny = rightBB.y - rightAA.y, lenz = rightBB.z - rightAA.z;
                const float rightS = 2.0 * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
                const float rightcost = rightS * (r - i);
                const float totalCost = rightcost + leftcost;
                if (totalCost < cost) {
                    cost = totalCost;
                    split = i;
                }
            }
            ccost[c] = cost, csplit[c] = split;
        });
        float cost = kINF;
        float split = l;
        for (int c = 0; c < nch; ++c)
            if (ccost[c] < cost) cost = ccost[c], split = csplit[c];
        *cost_out = cost, *split_out = split;
    }

    TNode* build_sah(int l, int r) {
        if (l > r) return nullptr;
        TNode* const nd = new_node(l, r);
        if ((r - l + 1) <= leaf_n) {
            nd->n = r - l + 1;
            nd->index = l;
            return nd;
        }
        float Cost = kINF;
        int Axis = 0;
        int Split = (l + r) / 2;
        const int cnt = r - l + 1;
        if (par_ok && cnt >= kParNodeMin) {
            {
                std::unique_ptr<vec3[]> lMax(new vec3[cnt]), lMin(new vec3[cnt]), rMax(new vec3[cnt]), rMin(new vec3[cnt]);
                for (int axis = 0; axis < 3; axis++) {
                    sort_axis(axis, l, r);
                    float cost, split;
                    sweep_chunked(l, r, lMax.get(), lMin.get(), rMax.get(), rMin.get(), &cost, &split);
                    if (cost < Cost) {
                        Cost = cost;
                        Axis = axis;
                        Split = split;
                    }
                }
            }
            sort_axis(Axis, l, r);
#pragma omp task default(shared)
            nd->left = build_sah(l, Split);
#pragma omp task default(shared)
            nd->right = build_sah(Split + 1, r);
#pragma omp taskwait
            return nd;
        }
        // prefix / suffix boxes of the sweep (c5:509-541) in this thread's scratch: no task scheduling point until they are
        // dead, so the buffer cannot be taken over by another node in between.  The reference fills them with -INF / INF and
        // folds in place; the running values below are the same recurrence with those initial values.
        // (the per-thread buffer only serves nodes up to kScratchTris triangles — 3 MB per worker thread, kept for the life
        // of the process; a larger node that ends up here, e.g. the root of a scene with a non-finite centroid, which cannot take
        // the all-threads path, gets a buffer of its own that is released when the node is done)
        constexpr size_t kScratchTris = 65536;
        static thread_local std::vector<vec3> scratch;
        std::unique_ptr<vec3[]> big;
        vec3* base;
        if ((size_t)cnt <= kScratchTris) {
            if (scratch.size() < 4 * (size_t)cnt) scratch.resize(4 * (size_t)cnt);
            base = scratch.data();
        } else {
            big.reset(new vec3[4 * (size_t)cnt]);
            base = big.get();
        }
        vec3* const leftMax = base;
        vec3* const leftMin = leftMax + cnt;
        vec3* const rightMax = leftMin + cnt;
        vec3* const rightMin = rightMax + cnt;
        for (int axis = 0; axis < 3; axis++) {
            sort_axis(axis, l, r);
            vec3 amx = v3(-kINF), amn = v3(kINF);
            for (int i = l; i <= r; i++) {
                amx = max3(amx, tmax[px[i].idx]), amn = min3(amn, tmin[px[i].idx]);
                leftMax[i - l] = amx, leftMin[i - l] = amn;
            }
            amx = v3(-kINF), amn = v3(kINF);
            for (int i = r; i >= l; i--) {
                amx = max3(amx, tmax[px[i].idx]), amn = min3(amn, tmin[px[i].idx]);
                rightMax[i - l] = amx, rightMin[i - l] = amn;
            }
            float cost = kINF;
            float split = l;  // `float split`, c5:545
            for (int i = l; i <= r - 1; ++i) {
                float lenx, leny, lenz;
                const vec3 leftAA = leftMin[i - l], leftBB = leftMax[i - l];
                lenx = leftBB.x - leftAA.x, leny = leftBB.y - leftAA.y, lenz = leftBB.z - leftAA.z;
                const float leftS = 2.0 * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
                const float leftcost = leftS * (i - l + 1);
                const vec3 rightAA = rightMin[i - l + 1], rightBB = rightMax[i - l + 1];
                lenx = rightBB.x - rightAA.x, leny = rightBB.y - rightAA.y, lenz = rightBB.z - rightAA.z;
                const float rightS = 2.0 * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
                const float rightcost = rightS * (r - i);
                const float totalCost = rightcost + leftcost;
                if (totalCost < cost) {
                    cost = totalCost;
                    split = i;
                }
            }
            if (cost < Cost) {
                Cost = cost;
                Axis = axis;
                Split = split;
            }
        }
        sort_axis(Axis, l, r);
        if (cnt >= kTaskMin) {
#pragma omp task default(shared)
            nd->left = build_sah(l, Split);
#pragma omp task default(shared)
            nd->right = build_sah(Split + 1, r);
#pragma omp taskwait
        } else {
            nd->left = build_sah(l, Split);
            nd->right = build_sah(Split + 1, r);
        }
        return nd;
    }

    TNode* build_median(int l, int r) {
        if (l > r) return nullptr;
        TNode* const nd = new_node(l, r);
        if ((r - l + 1) <= leaf_n) {
            nd->n = r - l + 1;
            nd->index = l;
            return nd;
        }
        const float lenx = nd->BB.x - nd->AA.x;
        const float leny = nd->BB.y - nd->AA.y;
        const float lenz = nd->BB.z - nd->AA.z;
        if (lenx >= leny && lenx >= lenz) sort_axis(0, l, r);
        if (leny >= lenx && leny >= lenz) sort_axis(1, l, r);
        if (lenz >= lenx && lenz >= leny) sort_axis(2, l, r);
        const int mid = (l + r) / 2;
        if (r - l + 1 >= kTaskMin) {
#pragma omp task default(shared)
            nd->left = build_median(l, mid);
#pragma omp task default(shared)
            nd->right = build_median(mid + 1, r);
#pragma omp taskwait
        } else {
            nd->left = build_median(l, mid);
            nd->right = build_median(mid + 1, r);
        }
        return nd;
    }
};

}  // namespace
}  // namespace rta

using namespace rta;

extern "C" int rt_bvh_build(rt_mesh* mesh, int leaf_n, int use_sah, float** nodes_out, int* n_nodes) {
    if (!mesh || !nodes_out || !n_nodes || leaf_n < 1) return set_error(RT_ERR_INVALID_ARG, "rt_bvh_build: bad arguments");
    const int nT = (int)mesh->tris.size();
    if (nT >= (1 << 24))
        return set_error(RT_ERR_INVALID_ARG, "rt_bvh_build: %d triangles: indices stored as fp32 are exact only below 2^24", nT);
    Builder b;
    b.leaf_n = leaf_n;
    b.px.resize(nT);
    b.tmin.resize(nT);
    b.tmax.resize(nT);
    bool all_finite = true;
#pragma omp parallel for schedule(static) reduction(&& : all_finite)
    for (int i = 0; i < nT; ++i) {
        const TriRec& t = mesh->tris[i];
        const vec3 c = (t.p1 + t.p2 + t.p3) / v3(3, 3, 3);
        all_finite = all_finite && fabsf(c.x) < INFINITY && fabsf(c.y) < INFINITY && fabsf(c.z) < INFINITY;  // (then p1, p2, p3 are finite too)
        b.px[i].cx = c.x, b.px[i].cy = c.y, b.px[i].cz = c.z, b.px[i].idx = i;
        b.tmin[i] = v3(fmin_(t.p1.x, fmin_(t.p2.x, t.p3.x)), fmin_(t.p1.y, fmin_(t.p2.y, t.p3.y)),
                       fmin_(t.p1.z, fmin_(t.p2.z, t.p3.z)));
        b.tmax[i] = v3(fmax_(t.p1.x, fmax_(t.p2.x, t.p3.x)), fmax_(t.p1.y, fmax_(t.p2.y, t.p3.y)),
                       fmax_(t.p1.z, fmax_(t.p2.z, t.p3.z)));
    }
    b.par_ok = all_finite;
    // dummy node 0 (c5:835-841); its fields are never read by a valid traversal.
    b.nodes.push_back(BNode());
    memset((void*)&b.nodes[0], 0, sizeof(BNode));
    TNode* root = nullptr;
#pragma omp parallel
#pragma omp single
    root = use_sah ? b.build_sah(0, nT - 1) : b.build_median(0, nT - 1);
    b.number(root);

    std::vector<TriRec> permuted((size_t)nT);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nT; ++i) permuted[i] = mesh->tris[b.px[i].idx];
    mesh->tris.swap(permuted);

    const int nN = (int)b.nodes.size();
    NodeRec* out = (NodeRec*)calloc((size_t)nN, sizeof(NodeRec));
    if (!out) return set_error(RT_ERR_OOM, "rt_bvh_build: out of memory");
    for (int i = 0; i < nN; ++i) {  // encode, c5:869-875
        out[i].left = (float)b.nodes[i].left, out[i].right = (float)b.nodes[i].right, out[i].pad0 = 0;
        out[i].n = (float)b.nodes[i].n, out[i].index = (float)b.nodes[i].index, out[i].pad1 = 0;
        out[i].AA = b.nodes[i].AA, out[i].BB = b.nodes[i].BB;
    }
    *nodes_out = (float*)out;
    *n_nodes = nN;
    return RT_OK;
}
