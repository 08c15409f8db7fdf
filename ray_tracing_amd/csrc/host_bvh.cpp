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
// four sorts stay sequential: each starts from the order the previous one left, as in the reference.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

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

struct Builder {
    std::vector<Proxy> px;
    std::vector<vec3> tmin, tmax;  // per original triangle: min/max over p1,p2,p3 as min(p1, min(p2, p3))
    std::vector<BNode> nodes;
    int leaf_n;

    TNode* new_node(int l, int r) {
        TNode* nd = new TNode();
        nd->left = nd->right = nullptr;
        nd->n = nd->index = 0;
        nd->AA = v3(1145141919, 1145141919, 1145141919);
        nd->BB = v3(-1145141919, -1145141919, -1145141919);
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
        if (axis == 0) std::sort(b, e, cmpx);
        if (axis == 1) std::sort(b, e, cmpy);
        if (axis == 2) std::sort(b, e, cmpz);
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
        std::vector<vec3> leftMax(cnt), leftMin(cnt), rightMax(cnt), rightMin(cnt);
        for (int axis = 0; axis < 3; axis++) {
            sort_axis(axis, l, r);
            std::fill(leftMax.begin(), leftMax.end(), v3(-kINF));
            std::fill(leftMin.begin(), leftMin.end(), v3(kINF));
            for (int i = l; i <= r; i++) {
                const vec3 mn = tmin[px[i].idx], mx = tmax[px[i].idx];
                const int bias = (i == l) ? 0 : 1;
                leftMax[i - l] = max3(leftMax[i - l - bias], mx);
                leftMin[i - l] = min3(leftMin[i - l - bias], mn);
            }
            std::fill(rightMax.begin(), rightMax.end(), v3(-kINF));
            std::fill(rightMin.begin(), rightMin.end(), v3(kINF));
            for (int i = r; i >= l; i--) {
                const vec3 mn = tmin[px[i].idx], mx = tmax[px[i].idx];
                const int bias = (i == r) ? 0 : 1;
                rightMax[i - l] = max3(rightMax[i - l + bias], mx);
                rightMin[i - l] = min3(rightMin[i - l + bias], mn);
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
        { std::vector<vec3>().swap(leftMax); std::vector<vec3>().swap(leftMin); }
        { std::vector<vec3>().swap(rightMax); std::vector<vec3>().swap(rightMin); }
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
    for (int i = 0; i < nT; ++i) {
        const TriRec& t = mesh->tris[i];
        const vec3 c = (t.p1 + t.p2 + t.p3) / v3(3, 3, 3);
        b.px[i].cx = c.x, b.px[i].cy = c.y, b.px[i].cz = c.z, b.px[i].idx = i;
        b.tmin[i] = v3(fmin_(t.p1.x, fmin_(t.p2.x, t.p3.x)), fmin_(t.p1.y, fmin_(t.p2.y, t.p3.y)),
                       fmin_(t.p1.z, fmin_(t.p2.z, t.p3.z)));
        b.tmax[i] = v3(fmax_(t.p1.x, fmax_(t.p2.x, t.p3.x)), fmax_(t.p1.y, fmax_(t.p2.y, t.p3.y)),
                       fmax_(t.p1.z, fmax_(t.p2.z, t.p3.z)));
    }
    // dummy node 0 (c5:835-841); its fields are never read by a valid traversal.
    b.nodes.push_back(BNode());
    memset((void*)&b.nodes[0], 0, sizeof(BNode));
    TNode* root = nullptr;
#pragma omp parallel
#pragma omp single
    root = use_sah ? b.build_sah(0, nT - 1) : b.build_median(0, nT - 1);
    b.number(root);

    std::vector<TriRec> permuted((size_t)nT);
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
