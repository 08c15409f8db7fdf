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

static int f2i_checked(float f, int* out) {
    if (!(f >= -1.0f && f < 16777216.0f)) return 0;
    const int i = (int)f;
    if ((float)i != f) return 0;
    *out = i;
    return 1;
}

// Writes the records of the entry BVH's subtree under internal node `id`, numbered in pre-order from `recno`; returns the
// largest triangle margin below it (each record carries its two children's margins, as the reference tree's records do).
struct EntryEmit {
    const FastBvhNode* fn;
    const int* perm;
    const float* tri_margin;
    float* rec2;
    float emit(int id, int recno) {
        const FastBvhNode& nd = fn[id];
        const int ch[2] = {nd.left, nd.right};
        const int rn[2] = {recno + 1, recno + 1 + (fn[ch[0]].count > 0 ? 0 : fn[ch[0]].first)};
        float cm[2] = {0.0f, 0.0f};
        int ce[2];
        for (int c = 0; c < 2; ++c) {
            const FastBvhNode& cn = fn[ch[c]];
            if (cn.count > 0) {
                for (int i = cn.first; i < cn.first + cn.count; ++i) cm[c] = fmaxf(cm[c], tri_margin[perm[i]]);
                ce[c] = pack_leaf(cn.first, cn.count);
            } else {
                ce[c] = rn[c];
            }
        }
        const bool l_int = fn[ch[0]].count == 0, r_int = fn[ch[1]].count == 0;
        if (l_int && r_int && nd.first >= 2048) {
#pragma omp task default(shared)
            cm[0] = emit(ch[0], rn[0]);
            cm[1] = emit(ch[1], rn[1]);
#pragma omp taskwait
        } else {
            if (l_int) cm[0] = emit(ch[0], rn[0]);
            if (r_int) cm[1] = emit(ch[1], rn[1]);
        }
        float* q = rec2 + (size_t)recno * 16;
        for (int c = 0; c < 2; ++c) {
            const FastBvhNode& cn = fn[ch[c]];
            memcpy(q + 6 * c, cn.AA, 12), memcpy(q + 6 * c + 3, cn.BB, 12);
            memcpy(q + 12 + c, &ce[c], 4);
            memcpy(q + 14 + c, &cm[c], 4);
        }
        return fmaxf(cm[0], cm[1]);
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
    int root_entry2 = 0, depth2 = 0;
    double cost_ref = 0.0, cost_entry = 0.0;
    bool has_entry = false;
    if (cull_eps < __builtin_inff() && !order.empty() && entry_mode >= 0 && (nT >= 1024 || entry_mode > 0)) {
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
                if (tri_margin[i] < 0.0f) continue;  // can never be hit
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
                    ttri2.alloc(perm.size() * 12);
#pragma omp parallel for schedule(static)
                    for (size_t k = 0; k < perm.size(); ++k) memcpy(&ttri2[k * 12], &ttri[(size_t)perm[k] * 12], 48);
                    PREP_MARK("  entry triangles");
                    // records in pre-order (same 64-B layout as the reference tree's): an internal node's `first` holds the
                    // number of internal nodes of its subtree, so the record numbers follow top-down — the node itself, then
                    // its left subtree, then its right one — and subtrees are emitted as parallel tasks
                    rec2.alloc((size_t)fn[root].first * 16);
                    EntryEmit em{fn.data(), perm.data(), tri_margin.data(), rec2.data()};
#pragma omp parallel
#pragma omp single
                    (void)em.emit(root, 0);
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
                    // from some level on it lies inside the box by more than mu = climb_eps + its shape margin on all six
                    // sides.  An accepted hit point is within a fraction of that margin of the triangle (culling lemma), so it
                    // is interior by ~mu, the slab parameters bracket t_W by mu / |d_i| — three orders of magnitude more than
                    // their rounding error — and the reference's hitAABB returns > 0 for certain.  Only the levels below need
                    // the explicit check; their number goes into bits 24..31 of leafx.  Valid for rays whose margin eps_ray
                    // <= climb_eps = 4 x the scene part (origins within 3 scene extents); other rays climb to the root.
                    PREP_MARK("  leafx / parentx");
                    climb_eps = 4.0f * cull_eps;
#pragma omp parallel for schedule(dynamic, 4096)
                    for (int i = 0; i < nT; ++i) {
                        if (tri_margin[i] < 0.0f) continue;
                        const float* q = &ttri[(size_t)i * 12];
                        double lo[3], hi[3];
                        for (int c = 0; c < 3; ++c) {
                            lo[c] = fmin((double)q[c], fmin((double)q[3 + c], (double)q[6 + c]));
                            hi[c] = fmax((double)q[c], fmax((double)q[3 + c], (double)q[6 + c]));
                        }
                        const double mu = (double)climb_eps + (double)tri_margin[i];
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
    out->climb_eps = climb_eps, out->root_entry2 = root_entry2, out->depth2 = depth2;
    out->cost_ref = cost_ref, out->cost_entry = cost_entry, out->has_entry = has_entry;
    return RT_OK;
}

}  // namespace rta
