// Host-side preparation of a scene for the device layer: everything rt_scene_create derives from the caller's arrays
// (Triangle_encoded[] c5:54-64, BVHNode_encoded[] c5:65-70, hdrMap / hdrCache) before anything touches the GPU — BVH
// validation and the 64-byte two-child node records, the 48-byte triangle records with the plane normal of f5:180, the
// per-triangle material tables, the culling margins and their preconditions, the entry BVH with the
// climb tables, the HDR texels.  Compiled by g++ with OpenMP: at 10 M triangles these loops are seconds on one core, and
// every start of the program pays them (the on-disk cache only saves the BVH build).  The per-triangle and per-record
// loops are independent per element, so they run on all cores with identical results.  -ffp-contract=off.
#include <math.h>
#include <string.h>

#include <utility>

#include "rt_host.h"
#include "rt_trav.h"

#ifdef RT_PREP_TIMING  // stage times on stderr (development builds only)
#include <chrono>
#define PREP_MARK(name)                                                                                          \
    do {                                                                                                          \
        const auto now_ = std::chrono::steady_clock::now();                                                      \
        fprintf(stderr, "prepare_scene: %-28s %7.3f s\n", name, std::chrono::duration<double>(now_ - mark_).count()); \
        mark_ = now_;                                                                                             \
    } while (0)
#define PREP_BEGIN() auto mark_ = std::chrono::steady_clock::now()
#else
#define PREP_MARK(name) ((void)0)
#define PREP_BEGIN() ((void)0)
#endif

namespace rta {

constexpr double kClimbDiv = 8.0;  // short climbs: mu = climb_eps / kClimbDiv + shape margin (see below)

static int f2i_checked(float f, int* out) {
    if (!(f >= -1.0f && f < 16777216.0f)) return 0;
    const int i = (int)f;
    if ((float)i != f) return 0;
    *out = i;
    return 1;
}

// The entry BVH reaches the device 4 wide: a walk on a scene that does not fit the caches is a chain of dependent memory
// round trips (one per node visited), and a 4-wide node does the work of two binary levels in one.  The binary tree of
// host_fastbvh.cpp is collapsed greedily (the child with the largest box is replaced by its own two children until there are
// four, or only leaves are left).  A record is 128 bytes = two halves in the layout of the reference tree's records
// (two boxes, two entries, two margins each), so the kernel tests children (0, 1) with the code it uses for the reference
// tree and children (2, 3) with the same code again; a missing child has entry kNoChild.
// measure(): per 4-wide node (named by the binary node it grew from) the number of 4-wide nodes in its subtree (records are
// numbered in pre-order, so a child's number follows from the sizes of the subtrees before it), the deepest stack a walk can
// need below it and the largest triangle margin below it.  emit(): writes the records.  Both as task-parallel recursions.
struct EntryEmit4 {
    const FastBvhNode* fn;
    const int* perm;
    const float* tri_margin;
    int* cnt4;      // per binary node: 4-wide nodes in the subtree of the 4-wide node grown from it
    float* margin;  // per binary node that is a 4-wide node: largest triangle margin below
    float* rec2 = nullptr;

    static double area(const FastBvhNode& n) {
        const double x = (double)n.BB[0] - n.AA[0], y = (double)n.BB[1] - n.AA[1], z = (double)n.BB[2] - n.AA[2];
        return (x < 0 || y < 0 || z < 0) ? 0.0 : 2.0 * (x * y + x * z + y * z);
    }
    int collapse(int id, int ch[4]) const {
        int k = 2;
        ch[0] = fn[id].left, ch[1] = fn[id].right;
        while (k < 4) {
            int pick = -1;
            double best = -1.0;
            for (int j = 0; j < k; ++j)
                if (fn[ch[j]].count == 0 && area(fn[ch[j]]) > best) best = area(fn[ch[j]]), pick = j;
            if (pick < 0) break;
            const int c = ch[pick];
            ch[pick] = fn[c].left, ch[k++] = fn[c].right;
        }
        return k;
    }
    float leaf_margin(const FastBvhNode& cn) const {
        float m = 0.0f;
        for (int i = cn.first; i < cn.first + cn.count; ++i) m = fmaxf(m, tri_margin[perm[i]]);
        return m;
    }
    // returns the stack entries a walk below `id` can hold at once (its own pending children included)
    int measure(int id) {
        int ch[4];
        const int k = collapse(id, ch);
        int need[4] = {0, 0, 0, 0};
        const bool par = fn[id].first >= 4096;  // (`first` of an internal binary node: internal nodes below it)
        for (int j = 0; j < k; ++j) {
            if (fn[ch[j]].count > 0) continue;
            if (par) {
#pragma omp task default(shared) firstprivate(j)
                need[j] = measure(ch[j]);
            } else {
                need[j] = measure(ch[j]);
            }
        }
#pragma omp taskwait
        int n4 = 1, deepest = 0;
        float m = 0.0f;
        for (int j = 0; j < k; ++j) {
            const FastBvhNode& cn = fn[ch[j]];
            if (cn.count > 0) {
                m = fmaxf(m, leaf_margin(cn));
            } else {
                n4 += cnt4[ch[j]], m = fmaxf(m, margin[ch[j]]);
                if (need[j] > deepest) deepest = need[j];
            }
        }
        cnt4[id] = n4, margin[id] = m;
        return (k - 1) + deepest;  // while one child's subtree is walked, the other k - 1 may be on the stack
    }
    void emit(int id, int recno) {
        int ch[4];
        const int k = collapse(id, ch);
        float* q = rec2 + (size_t)recno * 32;
        int next = recno + 1;
        const bool par = fn[id].first >= 4096;
        for (int j = 0; j < 4; ++j) {
            float* h = q + 16 * (j >> 1);  // the half record of children (0, 1) or (2, 3)
            const int c = j & 1;
            int e = kNoChild;
            float m = 0.0f;
            if (j < k) {
                const FastBvhNode& cn = fn[ch[j]];
                memcpy(h + 6 * c, cn.AA, 12), memcpy(h + 6 * c + 3, cn.BB, 12);
                if (cn.count > 0) {
                    e = pack_leaf(cn.first, cn.count), m = leaf_margin(cn);
                } else {
                    e = next, m = margin[ch[j]];
                    const int child = ch[j], child_rec = next;
                    next += cnt4[ch[j]];
                    if (par) {
#pragma omp task default(shared) firstprivate(child, child_rec)
                        emit(child, child_rec);
                    } else {
                        emit(child, child_rec);
                    }
                }
            } else {
                for (int t = 0; t < 6; ++t) h[6 * c + t] = 0.0f;
            }
            memcpy(h + 12 + c, &e, 4);
            memcpy(h + 14 + c, &m, 4);
        }
#pragma omp taskwait
    }
};

int prepare_scene(const float* tris, int nT, const float* nodes, int nN, const float* hdr_map, const float* hdr_cache, int hdrW,
                  int hdrH, int entry_mode, ScenePrep* out) {
    // ---- validate the BVH (root = node 1, f5:268) and number the internal nodes in pre-order -----------------
    PREP_BEGIN();
    const NodeRec* N = (const NodeRec*)nodes;
    std::vector<int> rec_of(nN > 0 ? nN : 0, -1);   // internal node -> record index
    std::vector<int> entry(nN > 0 ? nN : 0, 0);     // node -> packed entry
    std::vector<int> order;                        // internal nodes in record order
    int max_depth = 0;
    const bool has_root = nN >= 2;
    // per triangle: how many reachable leaves hold it (saturating at 2).  The reference only ever tests triangles of leaves it
    // reaches (f5:243-256), so a triangle of no leaf can never be hit, and one held by two leaves has no single "its leaf"
    std::vector<unsigned char> cover(nT > 0 ? nT : 0, 0);
    bool leaves_overlap = false;
    if (has_root) {
        std::vector<char> seen(nN, 0);
        std::vector<std::pair<int, int>> stack;  // node, depth
        stack.push_back({1, 0});
        while (!stack.empty()) {
            const int id = stack.back().first, depth = stack.back().second;
            stack.pop_back();
            if (id < 1 || id >= nN) return set_error(RT_ERR_BAD_BVH, "rt_scene_create: node index %d out of range [1,%d)", id, nN);
            if (seen[id]) return set_error(RT_ERR_BAD_BVH, "rt_scene_create: node %d reachable twice (not a tree)", id);
            seen[id] = 1;
            if (depth > max_depth) max_depth = depth;
            int l, r, n, index;
            if (!f2i_checked(N[id].left, &l) || !f2i_checked(N[id].right, &r) || !f2i_checked(N[id].n, &n) ||
                !f2i_checked(N[id].index, &index))
                return set_error(RT_ERR_BAD_BVH, "rt_scene_create: node %d holds a non-integral index field", id);
            if (n > 0) {  // leaf, f5:274
                if (n > 128) return set_error(RT_ERR_BAD_BVH, "rt_scene_create: leaf %d holds %d triangles (max 128)", id, n);
                if (index < 0 || (long long)index + n > nT)
                    return set_error(RT_ERR_BAD_BVH, "rt_scene_create: leaf %d range [%d,%d) outside the %d triangles", id, index, index + n, nT);
                entry[id] = pack_leaf(index, n);
                for (int i = index; i < index + n; ++i) {
                    if (cover[i]) leaves_overlap = true; else cover[i] = 1;
                }
            } else {
                if (l <= 0 || r <= 0)
                    return set_error(RT_ERR_BAD_BVH, "rt_scene_create: internal node %d lacks a child (left=%d right=%d)", id, l, r);
                rec_of[id] = (int)order.size();
                entry[id] = rec_of[id];
                order.push_back(id);
                stack.push_back({r, depth + 1});  // left is numbered first (popped first)
                stack.push_back({l, depth + 1});
            }
        }
    }
    const int need_stack = max_depth + 1;
    if (need_stack > kMaxStack)
        return set_error(RT_ERR_BAD_BVH, "rt_scene_create: tree depth %d overflows the reference's stack[256]", max_depth);

    PREP_MARK("validate + number");
    // ---- repack ------------------------------------------------------------------------------------------------
    RawBuf<float> rec, ttri, dmat;  // (every element is written below: no zero fill, first touch by the thread that fills it)
    rec.alloc(order.size() * 16);
#pragma omp parallel for schedule(static)
    for (size_t k = 0; k < order.size(); ++k) {
        const NodeRec& nd = N[order[k]];
        const int l = (int)nd.left, r = (int)nd.right;
        float* q = &rec[k * 16];
        q[0] = N[l].AA.x, q[1] = N[l].AA.y, q[2] = N[l].AA.z, q[3] = N[l].BB.x, q[4] = N[l].BB.y, q[5] = N[l].BB.z;
        q[6] = N[r].AA.x, q[7] = N[r].AA.y, q[8] = N[r].AA.z, q[9] = N[r].BB.x, q[10] = N[r].BB.y, q[11] = N[r].BB.z;
        memcpy(&q[12], &entry[l], 4);
        memcpy(&q[13], &entry[r], 4);
    }
    ttri.alloc((size_t)nT * 12);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nT; ++i) {
        const TriRec& t = ((const TriRec*)tris)[i];
        const vec3 Nn = normalize(cross(t.p2 - t.p1, t.p3 - t.p1));  // f5:180
        float* q = &ttri[(size_t)i * 12];
        q[0] = t.p1.x, q[1] = t.p1.y, q[2] = t.p1.z, q[3] = t.p2.x, q[4] = t.p2.y, q[5] = t.p2.z;
        q[6] = t.p3.x, q[7] = t.p3.y, q[8] = t.p3.z, q[9] = Nn.x, q[10] = Nn.y, q[11] = Nn.z;
    }
    dmat.alloc((size_t)nT * kDmatVec * 4);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nT; ++i) {
        const rt_material& rm = ((const TriRec*)tris)[i].m;
        Mat m;
        m.emissive = v3(rm.emissive[0], rm.emissive[1], rm.emissive[2]);
        m.baseColor = v3(rm.baseColor[0], rm.baseColor[1], rm.baseColor[2]);
        m.subsurface = rm.subsurface, m.metallic = rm.metallic, m.specular = rm.specular;
        m.specularTint = rm.specularTint, m.roughness = rm.roughness, m.anisotropic = rm.anisotropic;
        m.sheen = rm.sheen, m.sheenTint = rm.sheenTint, m.clearcoat = rm.clearcoat;
        m.clearcoatGloss = rm.clearcoatGloss;
        const MatPre p = mat_precompute(m);
        float* q = &dmat[(size_t)i * kDmatVec * 4];
        q[16] = rm.baseColor[0], q[17] = rm.baseColor[1], q[18] = rm.baseColor[2], q[19] = rm.subsurface;
        q[20] = rm.metallic, q[21] = rm.roughness, q[22] = rm.sheen, q[23] = rm.clearcoat;
        q[0] = p.Cspec0.x, q[1] = p.Cspec0.y, q[2] = p.Cspec0.z, q[3] = p.Csheen.x, q[4] = p.Csheen.y, q[5] = p.Csheen.z;
        q[6] = p.a2, q[7] = p.a2m1, q[8] = p.g1_a, q[9] = p.g1_a2, q[10] = p.g1_a2m1, q[11] = p.g1_k;
        q[12] = p.p_d, q[13] = p.p_s, q[14] = p.p_c, q[15] = p.g1_lg;
    }

    PREP_MARK("records + triangles + materials");
    // ---- culling margins (hitAABB_cull) ------------------------------------------------------------------------------
    // Culling needs nested boxes.  A ray's margin scales with the scene extent; each box adds the shape term of the worst
    // triangle below it: 2^-14 * longest edge / sin(smallest angle / 2) (the point S + d*t that hitTriangle accepts can lie
    // that far beyond the tip of a needle-shaped triangle), kept per child in the node record.
    // "Acceptable" triangles are those hitTriangle can ever accept (finite vertices, finite non-zero plane normal); the
    // others never produce a hit and are ignored here.
    double cull_extent = 0.0;
    float cull_geo_max = 0.0f;
    bool cull_ok = has_root;
    std::vector<float> tri_margin(nT > 0 ? nT : 0, -1.0f);  // -1: not acceptable
    {
        const float inf = __builtin_inff();
        auto fin = [inf](float v) { return fabsf(v) < inf; };
#pragma omp parallel for schedule(static) reduction(max : cull_extent)
        for (int i = 0; i < nT; ++i) {
            const float* q = &ttri[(size_t)i * 12];
            bool ok = true;
            for (int k = 0; k < 12; ++k) ok = ok && fin(q[k]);
            ok = ok && (q[9] != 0.0f || q[10] != 0.0f || q[11] != 0.0f);
            if (!ok) continue;
            for (int k = 0; k < 9; ++k)
                if (fabs((double)q[k]) > cull_extent) cull_extent = fabs((double)q[k]);
            tri_margin[i] = tri_cull_margin(q);
        }
        // per node: the largest triangle margin below it (children come after their parent in `order`: walk it backwards)
        std::vector<float> node_margin(nN > 0 ? nN : 0, 0.0f);
        auto leaf_margin = [&](const NodeRec& cn) {
            float m = 0.0f;
            const int first = (int)cn.index, cnt = (int)cn.n;
            for (int i = first; i < first + cnt; ++i)
                if (tri_margin[i] > m) m = tri_margin[i];
            return m;
        };
        for (size_t k = order.size(); k-- > 0;) {
            const NodeRec& nd = N[order[k]];
            const int ch[2] = {(int)nd.left, (int)nd.right};
            float m = 0.0f;
            for (int c = 0; c < 2; ++c) {
                const NodeRec& cn = N[ch[c]];
                if (cn.n > 0.0f) node_margin[ch[c]] = leaf_margin(cn);
                if (node_margin[ch[c]] > m) m = node_margin[ch[c]];
                memcpy(&rec[k * 16 + 14 + c], &node_margin[ch[c]], 4);
                if (node_margin[ch[c]] < inf && node_margin[ch[c]] > cull_geo_max) cull_geo_max = node_margin[ch[c]];
            }
            node_margin[order[k]] = m;
        }
        // every box the traversal tests (all nodes but the root) must contain its children's boxes / its triangles
        bool nested = true;
#pragma omp parallel for schedule(static) reduction(&& : nested) reduction(max : cull_extent)
        for (size_t k = 0; k < order.size(); ++k) {
            const NodeRec& nd = N[order[k]];
            const int ch[2] = {(int)nd.left, (int)nd.right};
            bool ok = true;
            for (int c = 0; c < 2; ++c) {
                const NodeRec& cn = N[ch[c]];
                const float bb[6] = {cn.AA.x, cn.AA.y, cn.AA.z, cn.BB.x, cn.BB.y, cn.BB.z};
                for (int j = 0; j < 6; ++j) {
                    ok = ok && fin(bb[j]);
                    if (fabs((double)bb[j]) > cull_extent) cull_extent = fabs((double)bb[j]);
                }
                if (order[k] != 1)  // (the root's own box is never tested, f5:268-291)
                    ok = ok && cn.AA.x >= nd.AA.x && cn.AA.y >= nd.AA.y && cn.AA.z >= nd.AA.z && cn.BB.x <= nd.BB.x &&
                         cn.BB.y <= nd.BB.y && cn.BB.z <= nd.BB.z;
                if (cn.n > 0.0f) {  // leaf: its acceptable triangles lie inside its box
                    const int first = (int)cn.index, cnt = (int)cn.n;
                    for (int i = first; i < first + cnt; ++i) {
                        if (tri_margin[i] < 0.0f) continue;
                        const float* q = &ttri[(size_t)i * 12];
                        for (int v = 0; v < 3; ++v)
                            ok = ok && q[v * 3] >= cn.AA.x && q[v * 3] <= cn.BB.x && q[v * 3 + 1] >= cn.AA.y &&
                                 q[v * 3 + 1] <= cn.BB.y && q[v * 3 + 2] >= cn.AA.z && q[v * 3 + 2] <= cn.BB.z;
                    }
                }
            }
            nested = nested && ok;
        }
        cull_ok = cull_ok && nested;
    }
    PREP_MARK("margins + nesting");
    float cull_eps = __builtin_inff();
    if (cull_ok && cull_extent > 1e-10 && cull_extent < 1e30) cull_eps = nextafterf((float)(cull_extent / 8192.0), __builtin_inff());  // 2^-13 * extent

    // ---- entry BVH (host_fastbvh.cpp): only where the reference tree is expensive to walk ---------------------------------
    // Expected internal nodes a random line through the root box visits = sum over internal nodes of area / root area.
    // The reference's INF = 114514 median splits (c5:500-502) push that into the hundreds on large scenes; a binned-SAH tree
    // over the same triangles stays in the tens.  The path loop then walks the entry BVH first (conservative box test, real
    // triangle test) to get a distance bound or to prove there is nothing to hit, and the reference tree only inside that
    // bound (rt_device.hip, "phase").  Built when the reference tree costs > 32 and kept when at least twice as cheap.
    RawBuf<float> rec2, ttri2;
    std::vector<int> map2, parentx;
    std::vector<unsigned> leafx;
    float climb_eps = 0.0f;
    int root_entry2 = 0, depth2 = 0, stack2 = 0;  // stack2: stack entries a walk of the 4-wide entry BVH can hold
    double cost_ref = 0.0, cost_entry = 0.0;
    bool has_entry = false;
    // (not for a BVH whose leaves share triangles: the climb starts from THE leaf of the entry walk's hit)
    if (cull_eps < __builtin_inff() && !order.empty() && entry_mode >= 0 && (nT >= 1024 || entry_mode > 0) && !leaves_overlap) {
        auto area = [](const float* lo, const float* hi) {
            const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
            return (x < 0 || y < 0 || z < 0) ? 0.0 : 2.0 * (x * y + x * z + y * z);
        };
        const double root_area = area(&N[1].AA.x, &N[1].BB.x);
        for (size_t k = 0; k < order.size(); ++k) cost_ref += area(&N[order[k]].AA.x, &N[order[k]].BB.x);
        cost_ref = root_area > 0.0 ? cost_ref / root_area : 0.0;
        if (cost_ref > 32.0 || entry_mode > 0) {
            RawBuf<float> tb;
            tb.alloc((size_t)nT * 6);
            std::vector<char> use(nT, 0);
#pragma omp parallel for schedule(static)
            for (int i = 0; i < nT; ++i) {
                if (tri_margin[i] < 0.0f || !cover[i]) continue;  // can never be hit: not acceptable, or in no leaf the reference reaches
                use[i] = 1;
                const float* q = &ttri[(size_t)i * 12];
                for (int c = 0; c < 3; ++c) {
                    tb[(size_t)i * 6 + c] = fminf(q[c], fminf(q[3 + c], q[6 + c]));
                    tb[(size_t)i * 6 + 3 + c] = fmaxf(q[c], fmaxf(q[3 + c], q[6 + c]));
                }
            }
            PREP_MARK("  cost + triangle boxes");
            RawBuf<FastBvhNode> fn;
            int n_fn = 0;
            double sum_area = 0.0;
            std::vector<int> perm;
            const int root = build_fast_bvh(tb.data(), use.data(), nT, fn, &n_fn, perm, &depth2, &sum_area);
            PREP_MARK("  build_fast_bvh");
            if (root >= 0 && fn[root].count == 0) {
                const double ra2 = area(fn[root].AA, fn[root].BB);
                cost_entry = ra2 > 0.0 ? sum_area / ra2 : 0.0;
                if ((cost_entry * 2.0 < cost_ref || entry_mode > 0) && depth2 + 1 <= kMaxStack) {
                    // 4-wide records in pre-order (EntryEmit4 above)
                    std::vector<int> cnt4((size_t)n_fn, 0);
                    RawBuf<float> margin4;
                    margin4.alloc((size_t)n_fn);
                    EntryEmit4 em{fn.data(), perm.data(), tri_margin.data(), cnt4.data(), margin4.data()};
                    int need4 = 0;
#pragma omp parallel
#pragma omp single
                    need4 = em.measure(root);
                    if (need4 + 1 <= kMaxStack) {  // (always, below 2^24 triangles; otherwise the reference tree serves alone)
                    stack2 = need4 + 1;
                    ttri2.alloc(perm.size() * 12);
#pragma omp parallel for schedule(static)
                    for (size_t k = 0; k < perm.size(); ++k) memcpy(&ttri2[k * 12], &ttri[(size_t)perm[k] * 12], 48);
                    PREP_MARK("  entry triangles");
                    rec2.alloc((size_t)cnt4[root] * 32);
                    em.rec2 = rec2.data();
#pragma omp parallel
#pragma omp single
                    em.emit(root, 0);
                    PREP_MARK("  entry records + margins");
                    root_entry2 = 0;  // record 0 = the root (pre-order)
                    has_entry = true;
                    map2.swap(perm);  // (perm is not used below: the leaf margins above were its last readers)
                    // where each reference box lives, for the climb from a triangle's leaf to the root (phase 1)
                    leafx.assign(nT, 0u), parentx.assign(order.size(), -1);
#pragma omp parallel for schedule(static)
                    for (size_t k = 0; k < order.size(); ++k) {
                        const NodeRec& nd = N[order[k]];
                        const int ch[2] = {(int)nd.left, (int)nd.right};
                        for (int c = 0; c < 2; ++c) {
                            const NodeRec& cn = N[ch[c]];
                            const int x = (int)(k << 1) | c;
                            if (cn.n > 0.0f) {
                                for (int i = (int)cn.index; i < (int)cn.index + (int)cn.n; ++i) leafx[i] = (unsigned)x;
                            } else {
                                parentx[rec_of[ch[c]]] = x;
                            }
                        }
                    }
                    // Short climbs.  Boxes nest, so going up from a triangle's leaf the triangle sits deeper and deeper inside:
                    // from some level on it lies inside the box by more than mu on all six sides.  An accepted hit point P is
                    // within delta <= 46u*M + 21u*L/sin(theta) of the triangle (culling lemma; u = 2^-24,
                    // M = max|S_i| + extent), so P is interior by mu - delta; the reference's slab parameters are off by at most
                    // 6u*M/|d_i|, so its hitAABB returns > 0 for certain once mu - delta > 6u*M.  mu = climb_eps / 8 + the
                    // triangle's shape margin = 256u*M_max + 1024u*L/sin(theta/2): five times what is needed for every ray with
                    // M <= M_max = 4 * extent, i.e. eps_ray <= climb_eps = 4 x the scene part (origins within 3 scene extents);
                    // other rays climb to the root.  Only the levels below need the explicit check; their number goes into
                    // bits 24..31 of leafx.
                    PREP_MARK("  leafx / parentx");
                    climb_eps = 4.0f * cull_eps;
#pragma omp parallel for schedule(dynamic, 4096)
                    for (int i = 0; i < nT; ++i) {
                        if (tri_margin[i] < 0.0f || !cover[i]) continue;
                        const float* q = &ttri[(size_t)i * 12];
                        double lo[3], hi[3];
                        for (int c = 0; c < 3; ++c) {
                            lo[c] = fmin((double)q[c], fmin((double)q[3 + c], (double)q[6 + c]));
                            hi[c] = fmax((double)q[c], fmax((double)q[3 + c], (double)q[6 + c]));
                        }
                        const double mu = (double)climb_eps / kClimbDiv + (double)tri_margin[i];
                        unsigned x = leafx[i];
                        int level = 0;
                        for (;;) {
                            const float* b = &rec[(size_t)(x >> 1) * 16 + 6 * (x & 1u)];
                            bool inside = true;
                            for (int c = 0; c < 3; ++c) inside = inside && (lo[c] - (double)b[c] >= mu) && ((double)b[3 + c] - hi[c] >= mu);
                            if (inside) break;
                            ++level;
                            const int up = parentx[x >> 1];
                            if (up < 0) break;
                            x = (unsigned)up;
                        }
                        leafx[i] |= (unsigned)(level < 255 ? level : 255) << 24;
                    }
#ifdef RT_PREP_TIMING
                    {
                        double sum = 0;
                        long hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        for (int i = 0; i < nT; ++i) sum += leafx[i] >> 24, hist[(leafx[i] >> 24) < 7 ? (leafx[i] >> 24) : 7]++;
                        fprintf(stderr, "prepare_scene: explicit climb levels per triangle: mean %.2f; 0:%ld 1:%ld 2:%ld 3:%ld 4:%ld 5:%ld 6:%ld 7+:%ld; 4-wide records %zu, stack %d\n",
                                sum / (nT > 0 ? nT : 1), hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], rec2.size() / 32, stack2);
                    }
#endif
                    }
                }
            }
        }
    }


    PREP_MARK("entry BVH + climb tables");
    // ---- HDR texels: RGB + hdrCache's pdf channel (f5:651) in .w; per texel the direction SampleHdr computes from the cache's
    // (x, y) channels (f5:615-624) ------------------------------------------------------------------------------------------
    const size_t npx = (size_t)(hdr_map ? hdrW : 0) * (size_t)(hdr_map ? hdrH : 0);
    if (hdr_map) {
        out->hdr4.alloc(npx * 4);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < npx; ++i) {
            float* q = &out->hdr4[i * 4];
            q[0] = hdr_map[i * 3], q[1] = hdr_map[i * 3 + 1], q[2] = hdr_map[i * 3 + 2];
            q[3] = hdr_cache ? hdr_cache[i * 3 + 2] : 0.0f;
        }
        if (hdr_cache) {
            out->cdir4.alloc(npx * 4);
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < npx; ++i) {
                const vec3 L = hdrDirection(hdr_cache[i * 3], hdr_cache[i * 3 + 1]);
                float* q = &out->cdir4[i * 4];
                q[0] = L.x, q[1] = L.y, q[2] = L.z, q[3] = 0.0f;
            }
        }
    }

    PREP_MARK("HDR");
    out->has_root = has_root, out->need_stack = need_stack, out->root_entry = has_root ? entry[1] : 0;
    out->rec.swap(rec), out->ttri.swap(ttri), out->dmat.swap(dmat);
    out->cull_eps = cull_eps, out->cull_geo_max = cull_geo_max;
    out->rec2.swap(rec2), out->ttri2.swap(ttri2), out->map2.swap(map2), out->parentx.swap(parentx), out->leafx.swap(leafx);
    out->climb_eps = climb_eps, out->root_entry2 = root_entry2, out->stack2 = stack2;
    out->cost_ref = cost_ref, out->cost_entry = cost_entry, out->has_entry = has_entry;
    return RT_OK;
}

}  // namespace rta
