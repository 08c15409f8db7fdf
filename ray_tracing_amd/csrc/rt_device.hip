// Device layer: scene upload/validation/repacking and the HIP kernels of the per-pixel path loop (gfx950).
//
// Replaces pass1.draw() of the reference (c5:755 -> fshader.fsh main()).  Every (pixel, frameCounter) sample of
// frame_begin .. frame_begin+frame_count-1 is an independent work item traced by whichever lane is free; the samples of a
// pixel are then folded in frame order (fold_kernel), so the fp32 running mean of f5:954-955 is reproduced exactly.
// BVH traversal visits nodes in the reference's order without culling (f5:259-311), on a device-private repack of the
// reference arrays:
//   * node record (64 B): both children's boxes + packed child entries -> one record per internal visit instead of the
//     reference's 3 x 48 B node fetches;
//   * traversal triangle (48 B): p1 p2 p3 + the plane normal normalize(cross(p2-p1,p3-p1)) precomputed with the
//     identical fp32 sequence (f5:180) -> no sqrt/div per test; vertex normals + material (the other 96 B of
//     Triangle_encoded) are fetched once per closest hit, not per improvement (f5:250-253).
// Per-lane traversal stacks live in LDS ([depth][lane] layout, conflict-free), with a global spill only for trees deeper
// than the LDS stack.  Environment-visibility rays (f5:863-869) only need isHit, so they stop at the first hit.
// No MFMA: nothing here is a dense contraction.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "rt_host.h"
#include "rt_shade.h"
#include "rt_trav.h"

namespace rta {

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return set_error(RT_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                               __FILE__, __LINE__);                                        \
    } while (0)

constexpr int kBlock = 256;      // primary_kernel: 16x16 pixels; each wave64 owns an 8x8 sub-tile
#ifndef RT_LDS_DEPTH
#define RT_LDS_DEPTH 18
#endif
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 5
#endif
constexpr int kLdsDepth = RT_LDS_DEPTH;    // per-lane stack entries kept in LDS

// Sobol direction numbers, f5:339 (data table: 8 dimensions x 32 bits)
__device__ __align__(16) const unsigned d_sobolV[8 * 32] = {
    2147483648u, 1073741824u, 536870912u, 268435456u, 134217728u, 67108864u, 33554432u, 16777216u, 8388608u, 4194304u, 2097152u, 1048576u, 524288u, 262144u, 131072u, 65536u, 32768u, 16384u, 8192u, 4096u, 2048u, 1024u, 512u, 256u, 128u, 64u, 32u, 16u, 8u, 4u, 2u, 1u,
    2147483648u, 3221225472u, 2684354560u, 4026531840u, 2281701376u, 3422552064u, 2852126720u, 4278190080u, 2155872256u, 3233808384u, 2694840320u, 4042260480u, 2290614272u, 3435921408u, 2863267840u, 4294901760u, 2147516416u, 3221274624u, 2684395520u, 4026593280u, 2281736192u, 3422604288u, 2852170240u, 4278255360u, 2155905152u, 3233857728u, 2694881440u, 4042322160u, 2290649224u, 3435973836u, 2863311530u, 4294967295u,
    2147483648u, 3221225472u, 1610612736u, 2415919104u, 3892314112u, 1543503872u, 2382364672u, 3305111552u, 1753219072u, 2629828608u, 3999268864u, 1435500544u, 2154299392u, 3231449088u, 1626210304u, 2421489664u, 3900735488u, 1556135936u, 2388680704u, 3314585600u, 1751705600u, 2627492864u, 4008611328u, 1431684352u, 2147543168u, 3221249216u, 1610649184u, 2415969680u, 3892340840u, 1543543964u, 2382425838u, 3305133397u,
    2147483648u, 3221225472u, 536870912u, 1342177280u, 4160749568u, 1946157056u, 2717908992u, 2466250752u, 3632267264u, 624951296u, 1507852288u, 3872391168u, 2013790208u, 3020685312u, 2181169152u, 3271884800u, 546275328u, 1363623936u, 4226424832u, 1977167872u, 2693105664u, 2437829632u, 3689389568u, 635137280u, 1484783744u, 3846176960u, 2044723232u, 3067084880u, 2148008184u, 3222012020u, 537002146u, 1342505107u,
    2147483648u, 1073741824u, 536870912u, 2952790016u, 4160749568u, 3690987520u, 2046820352u, 2634022912u, 1518338048u, 801112064u, 2707423232u, 4038066176u, 3666345984u, 1875116032u, 2170683392u, 1085997056u, 579305472u, 3016343552u, 4217741312u, 3719483392u, 2013407232u, 2617981952u, 1510979072u, 755882752u, 2726789248u, 4090085440u, 3680870432u, 1840435376u, 2147625208u, 1074478300u, 537900666u, 2953698205u,
    2147483648u, 1073741824u, 1610612736u, 805306368u, 2818572288u, 335544320u, 2113929216u, 3472883712u, 2290089984u, 3829399552u, 3059744768u, 1127219200u, 3089629184u, 4199809024u, 3567124480u, 1891565568u, 394297344u, 3988799488u, 920674304u, 4193267712u, 2950604800u, 3977188352u, 3250028032u, 129093376u, 2231568512u, 2963678272u, 4281226848u, 432124720u, 803643432u, 1633613396u, 2672665246u, 3170194367u,
    2147483648u, 3221225472u, 2684354560u, 3489660928u, 1476395008u, 2483027968u, 1040187392u, 3808428032u, 3196059648u, 599785472u, 505413632u, 4077912064u, 1182269440u, 1736704000u, 2017853440u, 2221342720u, 3329785856u, 2810494976u, 3628507136u, 1416089600u, 2658719744u, 864310272u, 3863387648u, 3076993792u, 553150080u, 272922560u, 4167467040u, 1148698640u, 1719673080u, 2009075780u, 2149644390u, 3222291575u,
    2147483648u, 1073741824u, 2684354560u, 1342177280u, 2281701376u, 1946157056u, 436207616u, 2566914048u, 2625634304u, 3208642560u, 2720006144u, 2098200576u, 111673344u, 2354315264u, 3464626176u, 4027383808u, 2886631424u, 3770826752u, 1691164672u, 3357462528u, 1993345024u, 3752330240u, 873073152u, 2870150400u, 1700563072u, 87021376u, 1097028000u, 1222351248u, 1560027592u, 2977959924u, 23268898u, 437609937u};

#ifdef RT_OP_CYCLES
// profiling builds only (-DRT_OP_CYCLES): s_memtime ticks spent in each kind of pass, summed over waves
__device__ unsigned long long g_opcycles[8];
#endif
#ifdef RT_SHADE_CYCLES
// profiling builds only (-DRT_SHADE_CYCLES): s_memtime ticks per step of the shade pass, summed over waves; [15] = passes
__device__ unsigned long long g_shadecycles[16];
#define SHADE_MARK(k_) do { const long long t_ = clock64(); scy[k_] += (unsigned long long)(t_ - scy_t); scy_t = t_; } while (0)
#else
#define SHADE_MARK(k_) do { } while (0)
#endif

// ---- device view of a scene ------------------------------------------------------------------------------------
struct DevScene {
    const float4* ttri;   // 3 x float4 per triangle: (p1.xyz,p2.x) (p2.yz,p3.xy) (p3.z,N.xyz)
    const float* tris;    // Triangle_encoded[nT], the caller's layout
    const float4* nodes;  // 4 x float4 per internal node: (lAA,lBB.x) (lBB.yz,rAA.xy) (rAA.z,rBB) (lentry,rentry,lmargin,rmargin)
    const float4* dmat;   // 6 x float4 per triangle: material-only terms of chapter 5's BRDF (MatPre, rt_shade.h) + its per-evaluation fields
    const float4* hdr;    // hdrMap as RGB + the pdf channel of hdrCache at the same texel (hdrPdf reads both, f5:638-660)
    const float4* cdir;   // per hdrCache texel: the direction SampleHdr returns for it (f5:615-624), xyz-
    int hdrW, hdrH;
    int root_entry;       // packed entry of node 1
    int has_root;
    float cull_eps;       // scene part of a ray's culling margin (see hitAABB_cull); +inf = this scene is never culled
    float cull_geo_max;   // largest finite per-box margin of the node records
    const float4* nodes2; // entry BVH, 4 wide: 8 x float4 per node = two halves in the layout of `nodes` (children 0,1 | 2,3; entry kNoChild = none) ...
    const float4* ttri2;  // ... and its triangles (same layout as `ttri`), in its own leaf order
    const int* map2;      // entry-BVH triangle -> index in the caller's order
    const unsigned* leafx; // per triangle (caller's order): where its leaf's box lives = (record << 1 | side), 24 bits; bits 24..31: how
                          // many boxes of the climb need an explicit check for rays whose margin is <= climb_eps (255 = all)
    float climb_eps;      // the ray margin (hitAABB_cull's eps_ray) the short climbs were computed for
    const int* parentx;   // per record of `nodes`: where that node's own box lives = (parent record << 1 | side); -1 = root
    int root_entry2;
    int has_entry;        // walk the entry BVH first (phase 0), then check its answer against the reference tree (phase 1)
};

// packed stack entry: >= 0 internal record index; < 0 leaf: bit31 | (n-1) << 24 | first triangle index

struct Stats {
    unsigned rays, internal, leaves, tris, updates;
};

struct RenderParams {
    DevScene sc;
    int width, height, max_bounce, accumulate;
    int frame_begin, frame_count;
    float eye[3];
    float cam[16];
    float* accum;
    float* samples;        // radiance samples of this launch, [frame - frame_begin][pixel][3]
    const float2* sobol;   // this launch's sobolVec2(frame + 1, bounce) (f5:362-365, 883): [frame - frame_begin][bounce 0..3]
    unsigned* queue;       // work queues, one per XCD's band of tiles (64 B apart): the next unit (tile, frame group) of the band
    unsigned long long* clk;  // [0]: longest wave lifetime in shader-clock ticks (s_memtime), max over launches since the last reset
    int unit_frames;       // frames per unit
    int* spill;            // global stack spill, [depth - kLdsDepth][thread]
    long long spill_stride;
    unsigned long long* stats;  // 6 counters (STATS builds)
    int stats_executed;    // STATS builds: 0 = count the reference's walk (no culling, env-visibility rays walked fully);
                           // 1 = count what the product kernel executes (culling, env-visibility rays stop at the first hit)
    int refill;            // lanes waiting for a handler that trigger a handler pass
    int leaf_lanes;        // lanes holding a leaf that trigger a cooperative leaf pass
    int wait_budget;       // lane-passes spent waiting for the shade pass that trigger it
    int32_t* hit_tri;      // primary-hit query outputs
    float* hit_dist;
    // two-kernel form of the path loop (first_kernel -> render_kernel<..., SEG>):
    float4* states;        // per unit (tile, frame group) a region of unit_frames x 64 path states (4 x float4 each), filled from its start
    int* unit_count;       // per unit: how many states first_kernel left there
    float2* cp;            // per pixel: the Cranley-Patterson shift (f5:367-383)
};

struct LaneStack {
    int* lds;              // &s_stack[threadIdx.x]; entry k at lds[k * lstride]
    int* spill;            // &spill[global thread]; entry k at spill[(k - kLdsDepth) * stride]
    long long stride;
    int lstride;           // threads of the workgroup (the LDS part is laid out [depth][thread])
};

template <bool SPILL>
__device__ __forceinline__ void stack_push(const LaneStack& st, int sp, int v) {
    if (!SPILL || sp < kLdsDepth)
        st.lds[sp * st.lstride] = v;
    else
        st.spill[(long long)(sp - kLdsDepth) * st.stride] = v;
}
template <bool SPILL>
__device__ __forceinline__ int stack_pop(const LaneStack& st, int sp) {
    if (!SPILL || sp < kLdsDepth) return st.lds[sp * st.lstride];
    return st.spill[(long long)(sp - kLdsDepth) * st.stride];
}

enum TieRule { TIE_SHADER = 0, TIE_CHAP2 = 1 };

// ---- T3/T4: hitArray + hitBVH, f5:243-311 (TIE_SHADER) / c2:451-513 (TIE_CHAP2) --------------------------------
// Returns the closest triangle index (-1 = miss) and its distance.  ANYHIT returns at the first accepted hit.
template <bool ANYHIT, int TIE, bool STATS, bool SPILL>
__device__ __forceinline__ int hitBVH(const DevScene& sc, const LaneStack& stk, vec3 S, vec3 d, float* dist, Stats& st, bool stats_executed = false) {
    const bool CULL = !ANYHIT && (!STATS || stats_executed);  // statistics builds walk exactly the reference's nodes unless asked for the executed walk
    float best = INF_;
    int best_tri = -1;
    int best_first = -1;
    if (STATS) st.rays++;
    if (!sc.has_root) {
        *dist = best;
        return -1;
    }
    const vec3 invdir = v3(1.0f) / d;
    const bool slow = !(finite3(invdir) && finite3(S));
    const float ceps = CULL ? cull_margin(sc.cull_eps, sc.cull_geo_max, S, invdir, slow) : 0.0f;
    int sp = 0;
    int cur = sc.root_entry;
    while (true) {
        if (cur >= 0) {
            if (STATS) st.internal++;
            const float4 q0 = sc.nodes[cur * 4 + 0], q1 = sc.nodes[cur * 4 + 1], q2 = sc.nodes[cur * 4 + 2],
                         q3 = sc.nodes[cur * 4 + 3];
            const vec3 lAA = v3(q0.x, q0.y, q0.z), lBB = v3(q0.w, q1.x, q1.y);
            const vec3 rAA = v3(q1.z, q1.w, q2.x), rBB = v3(q2.y, q2.z, q2.w);
            float d1, d2;
            if (CULL && !slow) {
                float c1, c2;
                d1 = hitAABB_cull(S, invdir, lAA, lBB, ceps, q3.z, &c1);
                d2 = hitAABB_cull(S, invdir, rAA, rBB, ceps, q3.w, &c2);
                // chapter 2 also accepts t == best from a leaf further right (c2:512): still t <= best < t0c
                if (c1 > best) d1 = -1.0f;
                if (c2 > best) d2 = -1.0f;
            } else {
                d1 = hitAABB(S, invdir, lAA, lBB);
                d2 = hitAABB(S, invdir, rAA, rBB);
            }
            const int le = __float_as_int(q3.x), re = __float_as_int(q3.y);
            if (d1 > 0.0f && d2 > 0.0f) {
                if (d1 < d2) {  // left first
                    stack_push<SPILL>(stk, sp++, re);
                    cur = le;
                } else {        // right first (ties included)
                    stack_push<SPILL>(stk, sp++, le);
                    cur = re;
                }
                continue;
            } else if (d1 > 0.0f) {
                cur = le;
                continue;
            } else if (d2 > 0.0f) {
                cur = re;
                continue;
            }
        } else {
            if (STATS) st.leaves++;
            const int first = cur & 0x00ffffff;
            const int n = ((cur >> 24) & 0x7f) + 1;
            float leaf_best = INF_;  // STATS only: hitArray restarts from INF in every leaf (f5:246)
            for (int i = first; i < first + n; ++i) {
                const float4 a = sc.ttri[i * 3 + 0], b = sc.ttri[i * 3 + 1], c = sc.ttri[i * 3 + 2];
                float t;
                if (STATS) st.tris++;
                // (statistics builds count every improvement inside a leaf, f5:250-253: no cut-off there)
                if (hitTriangleT(v3(a.x, a.y, a.z), v3(a.w, b.x, b.y), v3(b.z, b.w, c.x), v3(c.y, c.z, c.w), S, d, &t, STATS ? __builtin_inff() : best)) {
                    // shader (f5:250, 278): strict < in visit order => the first visited candidate keeps a tie.
                    // chap2 (c2:457, 512): minimum over all visited leaves; a tie goes to the right-most leaf
                    // (larger first index) and, inside a leaf, to the lowest index.
                    bool better;
                    if (TIE == TIE_CHAP2)
                        better = (t < best) || (t == best && first > best_first);
                    else
                        better = t < best;
                    if (STATS && t < leaf_best) {  // H = improvements inside hitArray (f5:250-253)
                        st.updates++;
                        leaf_best = t;
                    }
                    if (better) {
                        best = t;
                        best_tri = i;
                        best_first = first;
                        if (ANYHIT) {
                            *dist = best;
                            return best_tri;
                        }
                    }
                }
            }
        }
        if (sp == 0) break;
        cur = stack_pop<SPILL>(stk, --sp);
    }
    *dist = best;
    return best_tri;
}


// ---- hitBVH for a closest-hit ray on a scene with an entry BVH, one lane per ray (first_kernel) -----------------------------
// The three phases of render_kernel's ENTRY instantiations (see there and DESIGN.md section 5, "The entry BVH"), as a plain loop:
//   0: walk the 4-wide entry BVH with the conservative box test and the real hitTriangle -> W, the closest acceptable triangle
//      (none: the reference misses too), and whether another triangle shares its distance exactly;
//   1: no tie: check, with the reference's own hitAABB, the boxes from W's reference leaf upwards (only the levels that need it,
//      `leafx` bits 24..31): all pass => the reference returns W at t_W;
//   2: a tie, a refuted check or a ray that must not be culled: the reference's own (culled) walk, hitBVH above.
// Statistics builds count what is executed (they come here only for rt_render_stats_executed).
template <bool STATS, bool SPILL>
__device__ __forceinline__ int hitBVH_entry(const DevScene& sc, const LaneStack& stk, vec3 S, vec3 d, float* dist, Stats& st) {
    const vec3 invdir = v3(1.0f) / d;
    const bool slow = !(finite3(invdir) && finite3(S));
    const float ceps = cull_margin(sc.cull_eps, sc.cull_geo_max, S, invdir, slow);
    if (!(ceps < __builtin_inff())) return hitBVH<false, TIE_SHADER, STATS, SPILL>(sc, stk, S, d, dist, st, true);
    if (STATS) st.rays++;
    float best = INF_;
    int best_tri = -1;
    bool tie = false;
    int sp = 0;
    int cur = sc.root_entry2;
    while (true) {
        if (cur >= 0) {
            if (STATS) st.internal++;
            const float4* nb = sc.nodes2 + (size_t)cur * 8;
            const float4 q0 = nb[0], q1 = nb[1], q2 = nb[2], q3 = nb[3], q4 = nb[4], q5 = nb[5], q6 = nb[6], q7 = nb[7];
            float c1, c2, c3, c4, e1, e2, e3, e4;
            (void)hitAABB_cull(S, invdir, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), ceps, q3.z, &c1, &e1);
            (void)hitAABB_cull(S, invdir, v3(q1.z, q1.w, q2.x), v3(q2.y, q2.z, q2.w), ceps, q3.w, &c2, &e2);
            (void)hitAABB_cull(S, invdir, v3(q4.x, q4.y, q4.z), v3(q4.w, q5.x, q5.y), ceps, q7.z, &c3, &e3);
            (void)hitAABB_cull(S, invdir, v3(q5.z, q5.w, q6.x), v3(q6.y, q6.z, q6.w), ceps, q7.w, &c4, &e4);
            int n0 = __float_as_int(q3.x), n1 = __float_as_int(q3.y), n2 = __float_as_int(q7.x), n3 = __float_as_int(q7.y);
            float k0 = (c1 <= e1 && e1 >= 0.0f && c1 <= best) ? __builtin_fmaxf(c1, 1e-30f) : INF_;  // (children 0 and 1 always exist)
            float k1 = (c2 <= e2 && e2 >= 0.0f && c2 <= best) ? __builtin_fmaxf(c2, 1e-30f) : INF_;
            float k2 = (c3 <= e3 && e3 >= 0.0f && c3 <= best && n2 != kNoChild) ? __builtin_fmaxf(c3, 1e-30f) : INF_;
            float k3 = (c4 <= e4 && e4 >= 0.0f && c4 <= best && n3 != kNoChild) ? __builtin_fmaxf(c4, 1e-30f) : INF_;
#define RT_CSWAP(ka, kb, na, nb_)                    \
    {                                                \
        const bool sw_ = kb < ka;                    \
        const float tk_ = sw_ ? kb : ka;             \
        kb = sw_ ? ka : kb, ka = tk_;                \
        const int tn_ = sw_ ? nb_ : na;              \
        nb_ = sw_ ? na : nb_, na = tn_;              \
    }
            RT_CSWAP(k0, k1, n0, n1) RT_CSWAP(k2, k3, n2, n3) RT_CSWAP(k0, k2, n0, n2) RT_CSWAP(k1, k3, n1, n3) RT_CSWAP(k1, k2, n1, n2)
#undef RT_CSWAP
            if (k3 < INF_) stack_push<SPILL>(stk, sp++, n3);
            if (k2 < INF_) stack_push<SPILL>(stk, sp++, n2);
            if (k1 < INF_) stack_push<SPILL>(stk, sp++, n1);
            if (k0 < INF_) {
                cur = n0;
                continue;
            }
        } else {
            if (STATS) st.leaves++;
            const int first = cur & 0x00ffffff, n = ((cur >> 24) & 0x7f) + 1;
            float leaf_best = INF_;
            for (int i = first; i < first + n; ++i) {
                const float4 a = sc.ttri2[i * 3 + 0], b = sc.ttri2[i * 3 + 1], c = sc.ttri2[i * 3 + 2];
                float t;
                if (STATS) st.tris++;
                if (hitTriangleT(v3(a.x, a.y, a.z), v3(a.w, b.x, b.y), v3(b.z, b.w, c.x), v3(c.y, c.z, c.w), S, d, &t)) {
                    if (STATS && t < leaf_best) st.updates++, leaf_best = t;
                    if (t == best && i != best_tri) tie = true;  // a tie the entry BVH's order must not decide
                    if (t < best) best = t, best_tri = i, tie = false;
                }
            }
        }
        if (sp == 0) break;
        cur = stack_pop<SPILL>(stk, --sp);
    }
    if (best_tri < 0) {  // nothing acceptable on this ray: the reference misses too
        *dist = INF_;
        return -1;
    }
    if (!tie) {
        const int w = sc.map2[best_tri];
        const unsigned lx = sc.leafx[w];
        int steps = (ceps <= sc.climb_eps) ? (int)(lx >> 24) : 255;
        bool verified = steps == 0;
        int x = (int)(lx & 0x00ffffffu);
        while (!verified) {
            if (STATS) st.internal++;
            const int rec = x >> 1;
            const float4* nb = sc.nodes + (size_t)rec * 4;
            const float4 q0 = nb[0], q1 = nb[1], q2 = nb[2], q3 = nb[3];
            const int up = sc.parentx[rec];
            float c;
            const float dv = (x & 1) ? hitAABB_cull(S, invdir, v3(q1.z, q1.w, q2.x), v3(q2.y, q2.z, q2.w), ceps, q3.w, &c)
                                     : hitAABB_cull(S, invdir, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), ceps, q3.z, &c);
            if (!(dv > 0.0f)) break;  // refuted: the reference never reaches W's leaf
            if (steps != 255) --steps;
            if (up < 0 || steps == 0) verified = true; else x = up;
        }
        if (verified) {
            *dist = best;
            return w;
        }
    }
    if (STATS) st.rays--;  // (hitBVH counts the ray again)
    return hitBVH<false, TIE_SHADER, STATS, SPILL>(sc, stk, S, d, dist, st, true);
}

// ---- rest of hitTriangle for the closest hit only (f5:205-219; f4/f3:393-406) ----------------------------------
struct Hit {
    vec3 hitPoint, normal, viewDir;
    int tri;
};

template <int MODE>
__device__ __forceinline__ Hit finish_hit(const DevScene& sc, vec3 S, vec3 d, int tri, float t) {
    const float4 a = sc.ttri[tri * 3 + 0], b = sc.ttri[tri * 3 + 1], c = sc.ttri[tri * 3 + 2];
    const vec3 p1 = v3(a.x, a.y, a.z), p2 = v3(a.w, b.x, b.y), p3 = v3(b.z, b.w, c.x), N0 = v3(c.y, c.z, c.w);
    const float* T = sc.tris + (size_t)tri * 36;
    const vec3 n1 = v3(T[9], T[10], T[11]), n2 = v3(T[12], T[13], T[14]), n3 = v3(T[15], T[16], T[17]);
    const bool isInside = dot(N0, d) > 0.0f;
    const vec3 P = S + d * t;
    float alpha, beta;
    if (MODE == RT_MODE_CHAP5) {
        alpha = (-(P.x - p2.x) * (p3.y - p2.y) + (P.y - p2.y) * (p3.x - p2.x)) /
                (-(p1.x - p2.x) * (p3.y - p2.y) + (p1.y - p2.y) * (p3.x - p2.x) + 1e-7f);
        beta = (-(P.x - p3.x) * (p1.y - p3.y) + (P.y - p3.y) * (p1.x - p3.x)) /
               (-(p2.x - p3.x) * (p1.y - p3.y) + (p2.y - p3.y) * (p1.x - p3.x) + 1e-7f);
    } else {
        alpha = (-(P.x - p2.x) * (p3.y - p2.y) + (P.y - p2.y) * (p3.x - p2.x)) /
                (-(p1.x - p2.x - 0.00005f) * (p3.y - p2.y + 0.00005f) + (p1.y - p2.y + 0.00005f) * (p3.x - p2.x + 0.00005f));
        beta = (-(P.x - p3.x) * (p1.y - p3.y) + (P.y - p3.y) * (p1.x - p3.x)) /
               (-(p2.x - p3.x - 0.00005f) * (p1.y - p3.y + 0.00005f) + (p2.y - p3.y + 0.00005f) * (p1.x - p3.x + 0.00005f));
    }
    const float gama = 1.0f - alpha - beta;
    vec3 Nsmooth = alpha * n1 + beta * n2 + gama * n3;
    Nsmooth = normalize(Nsmooth);
    Hit h;
    h.hitPoint = P;
    h.normal = isInside ? (-Nsmooth) : Nsmooth;
    h.viewDir = d;
    h.tri = tri;
    return h;
}

__device__ __forceinline__ Mat load_material(const DevScene& sc, int tri) {
    const float* T = sc.tris + (size_t)tri * 36 + 18;
    Mat m;
    m.emissive = v3(T[0], T[1], T[2]);
    m.baseColor = v3(T[3], T[4], T[5]);
    m.subsurface = T[6], m.metallic = T[7], m.specular = T[8];
    m.specularTint = T[9], m.roughness = T[10], m.anisotropic = T[11];
    m.sheen = T[12], m.sheenTint = T[13], m.clearcoat = T[14];
    m.clearcoatGloss = T[15];
    return m;
}

__device__ __forceinline__ MatPre load_mat_pre(const DevScene& sc, int tri) {
    const float4 a = sc.dmat[tri * kDmatVec + 0], b = sc.dmat[tri * kDmatVec + 1], c = sc.dmat[tri * kDmatVec + 2], d = sc.dmat[tri * kDmatVec + 3];
    MatPre q;
    q.Cspec0 = v3(a.x, a.y, a.z), q.Csheen = v3(a.w, b.x, b.y);
    q.a2 = b.z, q.a2m1 = b.w;
    q.g1_a = c.x, q.g1_a2 = c.y, q.g1_a2m1 = c.z, q.g1_k = c.w;
    q.p_d = d.x, q.p_s = d.y, q.p_c = d.z, q.g1_lg = d.w;
    return q;
}
// the material fields chapter 5's BRDF still reads per evaluation (the rest is in MatPre)
__device__ __forceinline__ Mat load_material_eval5(const DevScene& sc, int tri) {
    const float4 e = sc.dmat[tri * kDmatVec + 4], f = sc.dmat[tri * kDmatVec + 5];  // two 16-B loads instead of eight dwords of Triangle_encoded
    Mat m;
    m.emissive = v3(0);
    m.baseColor = v3(e.x, e.y, e.z);
    m.subsurface = e.w, m.metallic = f.x, m.specular = 0.0f;
    m.specularTint = 0.0f, m.roughness = f.y, m.anisotropic = 0.0f;
    m.sheen = f.z, m.sheenTint = 0.0f, m.clearcoat = f.w;
    m.clearcoatGloss = 0.0f;
    return m;
}

__device__ __forceinline__ bool lane_pixel(const RenderParams& P, int* px, int* py) {
    // block = 16x16 pixels, wave w = 8x8 sub-tile (w&1, w>>1), lane l = (l&7, l>>3)
    const int tilesx = (P.width + 15) >> 4;
    const int bx = blockIdx.x % tilesx, by = blockIdx.x / tilesx;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    *px = bx * 16 + (w & 1) * 8 + (l & 7);
    *py = by * 16 + (w >> 1) * 8 + (l >> 3);
    return *px < P.width && *py < P.height;
}

__device__ __forceinline__ void flush_stats(const RenderParams& P, const Stats& st, unsigned long long paths) {
    unsigned long long v[6] = {paths, st.rays, st.internal, st.leaves, st.tris, st.updates};
    for (int k = 0; k < 6; ++k) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&P.stats[k], x);
    }
}

// ---- the path loop as a per-lane state machine with a wave-level scheduler ------------------------------------
// fshader.fsh main() (f5:926-988) + pathTracingImportanceSampling (f5:850-923) / pathTracing (f4:505-541, f3:378-416),
// cut at every hitBVH call.  A lane works on one sample (pixel, frame) at a time.  At any moment it waits for exactly
// one "op":
//   OP_INT   one internal-node step of hitBVH (two child boxes, f5:282-307)
//   OP_LEAF  the triangles of the leaf it holds (hitArray, f5:243-256), served by the cooperative leaf pass
//   OP_H     its ray came back: the shade pass (primary f5:940-951, env-visibility f5:869-895, BRDF ray f5:896-920 /
//            f4:525-538 / f3:391-412, told apart by `kind`), which ends by emitting the lane's next ray
// Each pass of the wave's single loop ballots the three populations, picks ONE op and executes it for every lane waiting
// on it (persistent lanes; parked lanes are refilled with their next ray by the shade pass).  Cheap traversal steps keep
// running while lanes accumulate at the expensive shade pass, so each shade pass is executed by many lanes, whatever
// frame, bounce or ray kind each lane is in.  Per lane the sequence of nodes, leaves, triangles, arithmetic and RNG
// calls is exactly the reference's.
enum LaneOp { OP_INT = 0, OP_LEAF = 1, OP_H = 2, OP_DONE = 3, OP_END = 4 /* transient: the walk of the current phase just ended */ };
enum RayKind { K_PRIMARY = 0, K_SHADOW = 1, K_BOUNCE = 2, K_NONE = 3 };

constexpr int kWaitBudget = 4000;  // lane-passes; of the order of (VALU cost of a shade pass) / (one lane's share of a traversal pass)
constexpr int kHandlerLanes = 48;  // run a handler once this many lanes wait for it (or nothing is traversing)

constexpr int kWave = 64;         // render_kernel: one wave64 per workgroup; work units are 8x8 pixel tiles x kUnitFrames frames
constexpr int kMaxLaunchFrames = 1024;  // bound of RT_AMD_CHUNK (sizes the per-launch Sobol table)
constexpr int kLaunchFrames = 64;  // frames per kernel launch (sample buffer = W x H x frames x 12 B = 1.6 GB at 1080p; a launch
                                   // ramps up and drains for ~0.5 ms whatever its length: 32 -> 64 frames = +2.8 %, 128: +4.2 %)
constexpr int kUnitFrames = 2;     // frames per work unit (one 8x8 tile x this many frames = 128 samples)
constexpr size_t kMaxSampleBytes = (size_t)12 << 30;  // per-launch workspace: samples (W x H x frames x 12 B) + in the two-kernel form the
                                                      // path states (64 B per sample slot): 1080p x 64 frames = 1.6 + 8.5 GB (288 GB of HBM)
constexpr int kMaxWidth = 65535, kMaxHeight = 32767;  // a pixel travels as (x | y << 16) in one int
constexpr int kOverlapFrames = 33;  // launches of fewer frames overlap their neighbours (two workspace slots, internal streams); config 5,
                                    // 1080p, ms per frame without / with overlap: 8 frames per call 0.428 / 0.365, 16: 0.377 / 0.357, 32: 0.355 / 0.344,
                                    // 64: 0.344 / 0.350 (tools/fpc_probe.py)
constexpr int kLeafLanes = 8;     // run a cooperative leaf pass once this many lanes hold a leaf (or nothing else can run)
constexpr int kHandlerLanesChap3 = 36;
constexpr size_t kTwoKernelMinSamples = (size_t)1 << 22;  // launches of fewer samples run the one-kernel form (scene_workspace)
constexpr int kLeafLanesBig = 16, kHandlerLanesBig = 24;  // the same two thresholds for scenes with an entry BVH (rt_scene_set_tuning)

template <bool SPILL>
__device__ __forceinline__ void wstack_push(int* lds, int* spill, long long stride, int sp, int v) {
    if (!SPILL || sp < kLdsDepth)
        lds[sp * kWave] = v;
    else
        spill[(long long)(sp - kLdsDepth) * stride] = v;
}
template <bool SPILL>
__device__ __forceinline__ int wstack_pop(const int* lds, const int* spill, long long stride, int sp) {
    if (!SPILL || sp < kLdsDepth) return lds[sp * kWave];
    return spill[(long long)(sp - kLdsDepth) * stride];
}

// Inclusive prefix sum over the 64 lanes of a wave (all lanes active): four DPP row shifts inside each 16-lane row, then
// the two row broadcasts of gfx9 (lane 15 of a row into the next row; lane 31 into rows 2 and 3).
__device__ __forceinline__ int wave_inclusive_sum(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}

// SEG: the two-kernel form.  first_kernel (below) has traced every sample's camera ray — coherent work: the 64 rays of an 8x8
// tile walk the same nodes — written the samples that ended there, and left, per unit, the states of the paths that go on.  This
// kernel's work items are then those states (a surface point with its first secondary ray, or K_NONE), not (pixel, frame) pairs:
// no K_PRIMARY rays, no camera-ray code, and the Cranley-Patterson shift comes from a per-pixel table instead of two registers.
template <int MODE, bool STATS, bool SPILL, bool ENTRY, bool SEG>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_EU))) render_kernel(const RenderParams P) {
    __shared__ int s_stack[kLdsDepth * kWave];        // per-lane traversal stacks, [depth][lane]
    __shared__ float4 s_ray[2 * kWave];                // per lane: the ray in flight (S.xyz, d.x) (d.yz, phase, primary-hit triangle of the path)
    __shared__ int s_item[kWave];                      // cooperative leaf pass: item -> (triangle << 6 | owner lane)
    __shared__ unsigned long long s_key[kWave];        // per owner: min over its items of (t bits << 32 | triangle)
    __shared__ float s_t[kWave];                       // STATS builds: per item hit distance (INF = miss)
    __shared__ int s_tie[ENTRY ? kWave : 1];           // entry-BVH walks: per owner, two of its items share the smallest t
    const DevScene& sc = P.sc;
    const int lane = threadIdx.x;
    const int tilesx = (P.width + 7) >> 3, n_tiles = tilesx * ((P.height + 7) >> 3);
    const int n_groups = (P.frame_count + P.unit_frames - 1) / P.unit_frames;  // frame groups; a unit = (tile, group)
    const int xcd = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;   // HW_REG_XCC_ID: which of the 8 XCDs runs this wave
    const size_t n_pixels = (size_t)P.width * P.height;
    int* const stk_lds = &s_stack[lane];
    int* const stk_spill = SPILL ? (P.spill + ((long long)blockIdx.x * kWave + lane)) : nullptr;
    const long long stk_stride = P.spill_stride;
    const long long wave_begin = clock64();  // s_memtime: shader-clock ticks
    Stats st = {0, 0, 0, 0, 0};
    unsigned long long paths = 0;
    unsigned dg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // STATS builds: passes / active-lane sums per op
    unsigned sg[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // STATS builds: lanes taking each step of the shade pass
#ifdef RT_SHADE_CYCLES
    unsigned long long scy[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long scy_t = 0;
#endif

    // lane state
    int op = OP_DONE;
    int pxy = 0;                    // the sample this lane works on: pixel (px | py << 16) ...
    // ... its frame, counted from frame_begin (bits 0-9: a launch has at most 1024), the bounce the path is in (bits 10-29) and
    // the kind of the ray in flight (bits 30-31) share one register: all three are shade-pass state
    unsigned fbk = 0;
#define FIDX_ ((int)(fbk & 1023u))
#define BOUNCE_ ((int)((fbk >> 10) & 0xfffffu))
#define KIND_ ((int)(fbk >> 30))
#define SET_KIND_(k_) (fbk = (fbk & 0x3fffffffu) | ((unsigned)(k_) << 30))
    float cpu = 0.0f, cpv = 0.0f;   // Cranley-Patterson shift of that pixel (f5:367-383)
    unsigned seed = 0;
    vec3 d = v3(0, 0, 1), invdir = v3(0);  // ray in flight.  Its origin is read back from the LDS ray table (s_ray) where a pass
                                           // needs it: three live registers fewer across the whole loop (1-2 % on config 5)
    bool slow = false;              // this ray needs the NaN-exact box test
    // statistics builds walk exactly the reference's nodes (algorithmic counts) unless asked for the executed walk
    const bool CULL = !STATS || P.stats_executed != 0;
    float ceps = __builtin_inff();  // culling margin of the ray in flight (hitAABB_cull); +inf = never cull
    // Which tree the ray in flight is walking (ENTRY instantiations; otherwise always 2).
    //   0: the entry BVH — a device-private binned-SAH tree over the same triangles, walked with the conservative box test
    //      (hitAABB_cull's inflated box: it never rejects a box that holds a triangle hitTriangle would accept), so it sees
    //      every acceptable triangle: no hit there = no hit for the reference, and its closest hit W at t_W is the closest
    //      acceptable hit overall;
    //   1: check W against the reference tree: the reference finds W iff every box from W's leaf up to (not including) the
    //      root passes the reference's own test for this ray (f5:282-307) — one box per internal-node pass, bottom-up.  All
    //      pass and W is the only triangle at t_W => W at t_W is the reference's answer (its candidates are a subset of the
    //      acceptable triangles and W is the minimum of the superset);
    //   2: the reference's own walk (no entry BVH; rays that must not be culled; a failed check; a tie at t_W, where the
    //      reference's visit order decides; env-visibility rays whose first hit failed the check).
    // Phases 1 / 2 alone decide results; the entry BVH only decides how much of the reference tree has to be walked.
    int phase = 2;
    bool tie = false;               // phase 0: another triangle has exactly the closest distance
    int cur = 0, sp = 0, best_tri = -1;  // cur: record of the internal node to visit, or (< 0) the leaf being scanned — a leaf entry
                                         // shrinks as its triangles are served: its first triangle is the next one to test
    float best = INF_;
    float leaf_best = INF_;         // STATS only
    vec3 hitP = v3(0), N = v3(0), V = v3(0);  // current surface point: hit.hitPoint, hit.normal, -hit.viewDir
    int surf_tri = -1;              // triangle of the current surface point (material source)
    // (the triangle of the path's primary hit — Le at the end of the sample, -1 = primary miss — is written and read once per
    // sample: it lives in the spare word of the lane's LDS ray-table entry, s_ray[2 * lane + 1].w, not in a register)
    vec3 Lo = v3(0), history = v3(1);

#ifdef RT_OP_CYCLES
    unsigned long long oc[4] = {0, 0, 0, 0}, acq[2] = {0, 0};
    const long long oc_begin = clock64();
#endif
    // The launch's samples (pixel, frame) are independent work items.  They are grouped in units (one 8x8 tile x
    // unit_frames frames) behind one global counter; a wave takes a unit, hands its items in order to lanes as they become
    // free, and takes the next unit while its last paths are still in flight — so no lane waits for a neighbour's longer
    // path and every wave runs until the queue is empty.  Samples go to the launch's sample buffer; fold_kernel then
    // applies f5:954-955 per pixel in frame order, which keeps the fp32 running mean exact.
    int unit_tx0 = 0, unit_ty0 = 0, unit_f0 = 0;  // wave-uniform: the unit being handed out (SEG: unit_f0 = its first state slot)
    int unit_items = 0, next_item = 0;            // its size and the first item not handed out yet
    bool pend_fin = false;          // the path ended in the BRDF-sampling step, after this pass's end-of-sample step: its sample is
                                    // stored (and the lane refilled) by the next shade pass
    bool queue_empty = false;
    unsigned open_bands = 0xffu;                  // wave-uniform: tile bands (one per XCD) that may still have units
    bool need_new = true;           // every lane asks for its first sample in the prologue pass
    int pick = OP_H;                // the prologue runs the shade pass once with no lane in OP_H: only its tail
    int waited = 0;                 // wave-uniform: lane-passes spent waiting for the shade pass since the last one
    for (;;) {
#ifdef RT_OP_CYCLES
        const long long oc_t0 = clock64();
        const int oc_pick = pick;
#endif
        if (pick >= OP_H) {
            // ---- the shade pass: every lane whose ray came back (primary / env-visibility / BRDF ray, told apart by
            // `kind`), step by step; a step runs once for all lanes that need it ------------------------------------
            bool to_B = false, to_C = false, fin = false, emit = false;
            const bool h = (op == OP_H) && !pend_fin;
            const bool isA = !SEG && h && KIND_ == K_PRIMARY, isD = h && KIND_ == K_BOUNCE;
            const bool isC = h && (KIND_ == K_SHADOW || KIND_ == K_NONE);
            const bool miss = best_tri < 0;
            const vec3 L = d;  // direction of the ray that came back
            vec3 S = v3(0);    // its origin (read back only where the hit record is finished); then the next ray's
            // step 1: BRDF value (and sampling pdf) for L at the current surface point: the BRDF ray always (f5:899-900,
            // f4:520, f3:396), the env-visibility ray when it was not occluded (f5:875-876)
            vec3 f_r = v3(0);
            float pdf_brdf = 0.0f;
            const bool need_eval = isD || (h && KIND_ == K_SHADOW && miss);
#ifdef RT_SHADE_CYCLES
            scy_t = clock64(), scy[8]++;
#endif
            if (STATS) { const int c_ = __popcll(__ballot(need_eval)); if (lane == 0) sg[0] += c_; }
            if (need_eval) {
                if (MODE == RT_MODE_CHAP5) {
                    const Mat mat = load_material_eval5(sc, surf_tri);
                    const MatPre pre = load_mat_pre(sc, surf_tri);
                    f_r = disney_eval5_pre(V, N, L, mat, pre, &pdf_brdf);
                } else if (MODE == RT_MODE_CHAP4) {
                    const Mat mat = load_material(sc, surf_tri);
                    vec3 tangent, bitangent;
                    getTagent(N, tangent, bitangent);
                    f_r = disney_eval<true, false>(V, N, L, tangent, bitangent, mat, nullptr);
                } else {
                    const Mat mat = load_material(sc, surf_tri);
                    f_r = mat.baseColor / PI_;
                }
            }
            // step 2: environment radiance (and its pdf) along L for every ray that escaped (f5:945, 873-874, 906-907)
            const bool d_alive = isD && (MODE != RT_MODE_CHAP5 || !(pdf_brdf <= 0.0f));  // f5:901
            const bool need_env = (isA && miss) || (h && KIND_ == K_SHADOW && miss) || (d_alive && miss);
            vec3 color = v3(0);
            float pdf_light = 0.0f;
            SHADE_MARK(0);  // BRDF eval
            if (STATS) { const int c_ = __popcll(__ballot(need_env)); if (lane == 0) sg[1] += c_; }
            if (need_env) {
                if (sc.hdr) {
                    float eu, ew;
                    toSphericalCoord(normalize(L), &eu, &ew);
                    const int texel = texel_index(ew, sc.hdrH) * sc.hdrW + texel_index(eu, sc.hdrW);
                    const float4 c = sc.hdr[texel];
                    color = v3(c.x, c.y, c.z);
                    if (MODE != RT_MODE_CHAP5) color = min3(color, v3(10));  // f4:162
                    // hdrPdf, f5:648-660: the pdf channel of hdrCache at the same texel rides in the map's spare .w (one gather)
                    if (MODE == RT_MODE_CHAP5 && !isA) pdf_light = hdrPdfConvert(c.w, ew, sc.hdrW);
                }
            }
            SHADE_MARK(1);  // env lookup
            // step 3: what each kind of ray adds
            if (isA && miss) {
                Lo = color, s_ray[2 * lane + 1].w = __int_as_float(-1), fin = true;
            }
            const float cosine_i = fmax_(0.0f, dot(N, L));  // chapters 3/4
            const float pdf_uniform = 1.0f / (2.0f * PI_);
            if (MODE == RT_MODE_CHAP5) {
                // f5:879-880 (env-visibility ray escaped), f5:910-911 (BRDF ray escaped), f5:915 + 920 (BRDF ray hit): three
                // disjoint sets of lanes, one shared quotient.  Each lane's numerator is built in its own line's order.
                const bool c_miss = h && KIND_ == K_SHADOW && miss, d_miss = d_alive && miss, d_hit = d_alive && !miss;
                if (isD && !d_alive) fin = true;  // f5:901
                if (c_miss || d_miss || d_hit) {
                    const float NdotL = dot(N, L);
                    vec3 base;
                    if (d_hit) {
                        const float* T = sc.tris + (size_t)best_tri * 36 + 18;
                        base = history * v3(T[0], T[1], T[2]);  // history * Le
                    } else {
                        const float mis_weight = misMixWeight(c_miss ? pdf_light : pdf_brdf, c_miss ? pdf_brdf : pdf_light);
                        base = mis_weight * history * color;
                    }
                    const vec3 num = c_miss ? (base * f_r * NdotL) : (base * NdotL * f_r);
                    Lo += num / (c_miss ? pdf_light : pdf_brdf);
                    if (d_hit) history *= NdotL * f_r / pdf_brdf;  // f5:920
                    if (d_miss) fin = true;
                }
            } else if (isD && miss) {
                Lo += history * color * f_r * cosine_i / pdf_uniform;  // f4:530, f3:401
                fin = true;
            }
            // step 4: the ray hit a surface: finish the hit record (smooth normal), move the path there
            SHADE_MARK(2);  // contributions
            if (STATS) { const int c_ = __popcll(__ballot((isA || d_alive) && !miss)); if (lane == 0) sg[2] += c_; }
            if ((isA || d_alive) && !miss) {
                {
                    const float4 sr = s_ray[2 * lane];
                    S = v3(sr.x, sr.y, sr.z);
                }
                const Hit nh = finish_hit<MODE>(sc, S, L, best_tri, best);
                if (isD) {
                    const float* T = sc.tris + (size_t)best_tri * 36 + 18;
                    const vec3 Le = v3(T[0], T[1], T[2]);
                    if (MODE != RT_MODE_CHAP5) {  // (chapter 5: done with the escaped rays above)
                        if (MODE == RT_MODE_CHAP4)
                            Lo += history * Le * cosine_i / pdf_uniform;        // f4:535: no f_r
                        else
                            Lo += history * Le * f_r * cosine_i / pdf_uniform;  // f3:407
                        history *= f_r * cosine_i / pdf_uniform;
                    }
                    fbk += 1u << 10;  // bounce++
                } else {
                    s_ray[2 * lane + 1].w = __int_as_float(best_tri);
                    Lo = v3(0), history = v3(1), fbk &= ~(0xfffffu << 10);  // bounce = 0
                }
                hitP = nh.hitPoint, N = nh.normal, V = -nh.viewDir, surf_tri = best_tri;
                if (BOUNCE_ < P.max_bounce) to_B = true; else fin = true;
            }
            // ---- end of a sample: Le + Li (f5:950) goes to the launch's sample buffer ----------------------------------
            SHADE_MARK(3);  // finish hit
            if (pend_fin) fin = true, pend_fin = false;
            if (STATS) { const int c_ = __popcll(__ballot(fin)); if (lane == 0) sg[5] += c_; }
            if (fin) {
                vec3 sample = Lo;
                const int first_tri = __float_as_int(s_ray[2 * lane + 1].w);
                if (first_tri >= 0) {
                    const float* T = sc.tris + (size_t)first_tri * 36 + 18;
                    sample = v3(T[0], T[1], T[2]) + Lo;
                }
                float* o = P.samples + ((size_t)FIDX_ * n_pixels + (size_t)(pxy >> 16) * P.width + (pxy & 0xffff)) * 3;
                o[0] = sample.x, o[1] = sample.y, o[2] = sample.z;
                if (STATS) paths++;
                need_new = true;
            }
            // ---- next work item: hand out the unit's remaining items in order, taking the next unit from the queue when it runs
            // out.  One-kernel form: items are the unit's (frame, pixel) pairs (pixels outside the image are skipped), followed by
            // seed, jitter, camera ray (f5:320-323, 932-937).  SEG: items are the path states first_kernel left in the unit's region --
            bool got = false;
            int slot = 0;
            for (;;) {
                const unsigned long long asking = __ballot(need_new && !got);
                if (asking == 0ull) break;  // wave-uniform
                if (next_item >= unit_items) {
                    if (queue_empty) break;
#ifdef RT_OP_CYCLES
                    const long long acq_t0 = clock64();
#endif
                    // take a unit: from this XCD's band of tiles first (neighbouring tiles walk the same nodes, and each
                    // XCD has its own L2), then from the next band that still has work
                    int tile = 0, group = 0;
                    bool have = false;
                    while (open_bands != 0u) {
                        const int q = (xcd + __builtin_ctz(((open_bands >> xcd) | (open_bands << (8 - xcd))) & 0xffu)) & 7;
                        const int band_begin = (int)((long long)n_tiles * q / 8), band_tiles = (int)((long long)n_tiles * (q + 1) / 8) - band_begin;
                        unsigned u = 0;
                        if (lane == 0) u = __hip_atomic_fetch_add(P.queue + q * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        u = (unsigned)__builtin_amdgcn_readfirstlane((int)u);
                        if (SEG) {
                            // (a unit in which no path goes on — every sample ended at its camera ray — is taken like any other and
                            // left at once; handing out only the non-empty ones through a compacted list was measured: the list's
                            // atomics cost 0.5 ms per launch and the path loop ran 40 % slower, not faster)
                            if (u < (unsigned)band_tiles * (unsigned)n_groups) {
                                const int t_ = band_begin + (int)(u % (unsigned)band_tiles), g_ = (int)(u / (unsigned)band_tiles);
                                tile = g_ * n_tiles + t_;  // first_kernel's block number
                                group = __builtin_amdgcn_readfirstlane(P.unit_count[tile]);
                                have = true;
                                break;
                            }
                        } else if (u < (unsigned)band_tiles * (unsigned)n_groups) {
                            tile = band_begin + (int)(u % (unsigned)band_tiles), group = (int)(u / (unsigned)band_tiles);
                            have = true;
                            break;
                        }
                        open_bands &= ~(1u << q);  // that band is handed out completely
                    }
                    if (!have) {
                        queue_empty = true;
                        break;
                    }
#ifdef RT_OP_CYCLES
                    acq[0] += (unsigned long long)(clock64() - acq_t0) + (unsigned long long)(unsigned)(group & 0), acq[1]++;  // (after the count arrived)
#endif
                    if (SEG) {
                        unit_f0 = tile * (P.unit_frames * kWave), unit_items = group, next_item = 0;  // (its region of `states`, its count)
                        if (unit_items == 0) continue;
                    } else {
                        unit_tx0 = (tile % tilesx) * 8, unit_ty0 = (tile / tilesx) * 8, unit_f0 = group * P.unit_frames;
                        const int nf = P.frame_count - unit_f0 < P.unit_frames ? P.frame_count - unit_f0 : P.unit_frames;
                        unit_items = nf * kWave, next_item = 0;
                    }
                }
                if (need_new && !got) {
                    const int it = next_item + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(asking >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)asking, 0u));
                    if (it < unit_items) {
                        if (SEG) {
                            slot = unit_f0 + it, got = true;
                        } else {
                            const int ix = unit_tx0 + (it & 7), iy = unit_ty0 + ((it >> 3) & 7);
                            pxy = ix | (iy << 16), fbk = (fbk & ~1023u) | (unsigned)(unit_f0 + (it >> 6));
                            got = ix < P.width && iy < P.height;
                        }
                    }
                }
                next_item += __popcll(asking);
            }
            if (need_new) {
                need_new = false;
                if (!got) {
                    op = OP_DONE;  // the queue is empty
                } else if (SEG) {
                    const float4* rec = P.states + (size_t)(unsigned)slot * 4;
                    const float4 ra = rec[0], rb = rec[1], rc = rec[2], re = rec[3];
                    hitP = v3(ra.x, ra.y, ra.z), N = v3(rb.z, rb.w, rc.x), V = v3(rc.y, rc.z, rc.w);
                    seed = __float_as_uint(re.x), pxy = __float_as_int(re.y), fbk = __float_as_uint(re.z), surf_tri = __float_as_int(re.w);
                    s_ray[2 * lane + 1].w = re.w;  // the primary hit = the first surface point
                    Lo = v3(0), history = v3(1);
                    if (KIND_ == K_NONE) {
                        to_C = true;  // no env-visibility ray for this point (f5:863): straight to BRDF sampling
                    } else {
                        S = hitP, d = v3(ra.w, rb.x, rb.y), emit = true;
                    }
                } else {
                    const float pixx = -1.0f + 2.0f * ((float)(pxy & 0xffff) + 0.5f) / (float)P.width;
                    const float pixy = -1.0f + 2.0f * ((float)(pxy >> 16) + 0.5f) / (float)P.height;
                    if (MODE == RT_MODE_CHAP5) cp_shift(pixx, pixy, P.width, P.height, &cpu, &cpv);
                    seed = pixel_seed(pixx, pixy, P.width, P.height, (unsigned)(P.frame_begin + FIDX_));
                    const float aax = (rand01(seed) - 0.5f) / (float)P.width;
                    const float aay = (rand01(seed) - 0.5f) / (float)P.height;
                    const float vx = pixx + aax, vy = pixy + aay, vz = -1.5f, vw = 0.0f;
                    const float* m = P.cam;  // column-major: m[col*4+row]
                    vec3 dir;
                    dir.x = (m[0] * vx + m[4] * vy) + (m[8] * vz + m[12] * vw);
                    dir.y = (m[1] * vx + m[5] * vy) + (m[9] * vz + m[13] * vw);
                    dir.z = (m[2] * vx + m[6] * vy) + (m[10] * vz + m[14] * vw);
                    S = v3(P.eye[0], P.eye[1], P.eye[2]), d = normalize(dir), SET_KIND_(K_PRIMARY), emit = true;
                }
            }
            // ---- start of a bounce (f5:856-866 env sample; chapters 3/4: the hemisphere sample) ------------------
            SHADE_MARK(6);  // end of sample + next work item (+ camera ray)
            if (STATS) { const int c_ = __popcll(__ballot(to_B)); if (lane == 0) sg[3] += c_; }
            if (to_B) {
                if (MODE == RT_MODE_CHAP5) {
                    const float r1 = rand01(seed), r2 = rand01(seed);
                    // SampleHdr (f5:612-625): the direction depends on the fetched texel alone, so it is tabulated per texel
                    // at scene creation (hdrDirection on the texel's x, y: the same fp32 operations)
                    const float4 ce = sc.cdir[texel_index(r2, sc.hdrH) * sc.hdrW + texel_index(r1, sc.hdrW)];
                    const vec3 Lh = v3(ce.x, ce.y, ce.z);
                    if (dot(N, Lh) > 0.0f) {
                        S = hitP, d = Lh, SET_KIND_(K_SHADOW), emit = true;
                    } else {
                        SET_KIND_(K_NONE), to_C = true;  // no visibility ray: straight to BRDF sampling
                    }
                } else {
                    const vec3 Lu = toNormalHemisphere(SampleHemisphereRand(seed), N);
                    S = hitP, d = Lu, SET_KIND_(K_BOUNCE), emit = true;
                }
            }
            // ---- BRDF importance sampling (f5:883-895): after the env-visibility ray, or at once when it was skipped ---
            SHADE_MARK(4);  // bounce start
            if (STATS) { const int c_ = __popcll(__ballot(isC || to_C)); if (lane == 0) sg[4] += c_; }
            if (MODE == RT_MODE_CHAP5 && (isC || to_C)) {
                const MatPre pre = load_mat_pre(sc, surf_tri);
                const float2 sb = P.sobol[FIDX_ * 4 + BOUNCE_];  // sobolVec2(frameCounter + 1, bounce): depends on (frame, bounce) only
                float su = cpu, sv = cpv;
                if (SEG) {
                    const float2 cs = P.cp[(size_t)(pxy >> 16) * P.width + (pxy & 0xffff)];
                    su = cs.x, sv = cs.y;
                }
                const float xi_1 = cp_rotate(sb.x, su);
                const float xi_2 = cp_rotate(sb.y, sv);
                const float xi_3 = rand01(seed);
                const vec3 Ls = SampleBRDF_pre(xi_1, xi_2, xi_3, V, N, pre);
                if (dot(N, Ls) <= 0.0f) {
                    pend_fin = true, op = OP_H;  // f5:891: the path ends; the next shade pass stores its sample
                } else {
                    S = hitP, d = Ls, SET_KIND_(K_BOUNCE), emit = true;
                }
            }
            SHADE_MARK(5);  // BRDF sampling
            // ---- hitBVH prologue (f5:260-268) --------------------------------------------------------------------
            if (STATS) { const int c_ = __popcll(__ballot(emit)); if (lane == 0) sg[6] += c_; }
            if (emit) {
                if (STATS) st.rays++;
                best = INF_, best_tri = -1, sp = 0;
                invdir = v3(1.0f) / d;
                slow = !(finite3(invdir) && finite3(S));
                ceps = CULL ? cull_margin(sc.cull_eps, sc.cull_geo_max, S, invdir, slow) : __builtin_inff();
                phase = (ENTRY && ceps < __builtin_inff()) ? 0 : 2;
                tie = false;
                cur = (ENTRY && phase == 0) ? sc.root_entry2 : sc.root_entry;
                s_ray[2 * lane] = make_float4(S.x, S.y, S.z, d.x);  // for the cooperative leaf pass
                s_ray[2 * lane + 1].x = d.y, s_ray[2 * lane + 1].y = d.z, s_ray[2 * lane + 1].z = __int_as_float(phase);  // (.w: the primary hit)
                if (!sc.has_root) {
                    op = OP_H;
                } else if (cur >= 0) {
                    op = OP_INT;
                } else {
                    op = OP_LEAF, leaf_best = INF_;
                    if (STATS) st.leaves++;
                }
            }
            SHADE_MARK(7);  // ray prologue
        } else if (pick == OP_INT) {
            // ---- one internal node (f5:282-307) ------------------------------------------------------------------
            if (op == OP_INT) {
                if (STATS) st.internal++;
                // phase 0 walks the 4-wide entry BVH: a record is two halves in the reference records' layout — children (0, 1)
                // in q0..q3, tested by the code below like the reference tree's two children, children (2, 3) in q4..q7
                const bool wide = ENTRY && phase == 0;
                const int rec = (ENTRY && phase == 1) ? (cur >> 1) : cur;  // phase 1: cur = (record << 1 | side) of the box to check
                const float4* nb = wide ? sc.nodes2 + rec * 8 : sc.nodes + rec * 4;
                const float4 q0 = nb[0], q1 = nb[1], q2 = nb[2];
                const float4 q3 = nb[3];  // the two child entries, the two children's culling margins
                int up = -1;  // phase 1: where the box above this one lives — requested with the record, not after its test
                if (ENTRY && phase == 1) up = sc.parentx[rec];
                const vec3 lAA = v3(q0.x, q0.y, q0.z), lBB = v3(q0.w, q1.x, q1.y);
                const vec3 rAA = v3(q1.z, q1.w, q2.x), rBB = v3(q2.y, q2.z, q2.w);
                // a child whose inflated box starts beyond the closest hit so far cannot change the result: skip it
                // (ceps = +inf for rays that must not be culled: then c1 = c2 = -inf)
                const float4 sr = s_ray[2 * lane];  // (written by this lane's own prologue; same wave: no barrier needed)
                const vec3 S = v3(sr.x, sr.y, sr.z);
                float c1, c2, e1, e2;
                float d1 = hitAABB_cull(S, invdir, lAA, lBB, ceps, q3.z, &c1, ENTRY ? &e1 : nullptr);
                float d2 = hitAABB_cull(S, invdir, rAA, rBB, ceps, q3.w, &c2, ENTRY ? &e2 : nullptr);
                bool verified = false, refuted = false;
                if (ENTRY && phase == 1) {
                    // the reference's own test of one box on the way up from W's leaf (never culled: ceps is finite here, so
                    // no operand is NaN and d1 / d2 are the ternary form's values)
                    const float dv = (cur & 1) ? d2 : d1;
                    if (dv > 0.0f) {
                        if (sp != 255) --sp;  // (sp is free during the climb: the explicit checks still to do)
                        if (up < 0 || sp == 0) verified = true; else cur = up;
                    } else {
                        refuted = true;
                    }
                    d1 = d2 = -1.0f;
                } else if (ENTRY && phase == 0) {  // entry BVH: the conservative test alone (hitAABB_entry), near child first by entry distance
                    d1 = (c1 <= e1 && e1 >= 0.0f && c1 <= best) ? __builtin_fmaxf(c1, 1e-30f) : -1.0f;
                    d2 = (c2 <= e2 && e2 >= 0.0f && c2 <= best) ? __builtin_fmaxf(c2, 1e-30f) : -1.0f;
                } else {
                    if (c1 > best) d1 = -1.0f;
                    if (c2 > best) d2 = -1.0f;
                }
                if (slow) {
                    d1 = hitAABB(S, invdir, lAA, lBB);
                    d2 = hitAABB(S, invdir, rAA, rBB);
                }
                const int le = __float_as_int(q3.x), re = __float_as_int(q3.y);
                bool popped = false;
                if (ENTRY && phase == 1) {
                    popped = true;  // (no descent: this lane climbs)
                    if (verified) {
                        op = OP_H, sp = 0;  // W at t_W is the reference's answer
                    } else if (refuted) {
                        phase = 2, sp = -1, op = OP_END;  // the reference never reaches W's leaf: its own walk decides (sp = -1: restart)
                    }
                } else if (wide) {
                    // the other two children, then all four in order of entry distance: the nearest is walked next, the
                    // others wait on the stack, nearest on top (the entry BVH never decides a result, only the order)
                    // (a record is one 128-byte line: the second half is an L1 hit by now, and loading it here keeps 16
                    // registers free during the first half's tests)
                    const float4 q4 = nb[4], q5 = nb[5], q6 = nb[6], q7 = nb[7];
                    float c3, c4, e3, e4;
                    (void)hitAABB_cull(S, invdir, v3(q4.x, q4.y, q4.z), v3(q4.w, q5.x, q5.y), ceps, q7.z, &c3, &e3);
                    (void)hitAABB_cull(S, invdir, v3(q5.z, q5.w, q6.x), v3(q6.y, q6.z, q6.w), ceps, q7.w, &c4, &e4);
                    int n0 = le, n1 = re, n2 = __float_as_int(q7.x), n3 = __float_as_int(q7.y);
                    float k0 = d1 > 0.0f ? d1 : INF_, k1 = d2 > 0.0f ? d2 : INF_;  // (children 0 and 1 always exist)
                    float k2 = (c3 <= e3 && e3 >= 0.0f && c3 <= best && n2 != kNoChild) ? __builtin_fmaxf(c3, 1e-30f) : INF_;
                    float k3 = (c4 <= e4 && e4 >= 0.0f && c4 <= best && n3 != kNoChild) ? __builtin_fmaxf(c4, 1e-30f) : INF_;
#define RT_CSWAP(ka, kb, na, nb_)                    \
    {                                                \
        const bool sw_ = kb < ka;                    \
        const float tk_ = sw_ ? kb : ka;             \
        kb = sw_ ? ka : kb, ka = tk_;                \
        const int tn_ = sw_ ? nb_ : na;              \
        nb_ = sw_ ? na : nb_, na = tn_;              \
    }
                    RT_CSWAP(k0, k1, n0, n1) RT_CSWAP(k2, k3, n2, n3) RT_CSWAP(k0, k2, n0, n2) RT_CSWAP(k1, k3, n1, n3) RT_CSWAP(k1, k2, n1, n2)
#undef RT_CSWAP
                    if (k3 < INF_) wstack_push<SPILL>(stk_lds, stk_spill, stk_stride, sp++, n3);
                    if (k2 < INF_) wstack_push<SPILL>(stk_lds, stk_spill, stk_stride, sp++, n2);
                    if (k1 < INF_) wstack_push<SPILL>(stk_lds, stk_spill, stk_stride, sp++, n1);
                    if (k0 < INF_) {
                        cur = n0;
                    } else if (sp == 0) {
                        op = OP_END;
                        popped = true;
                    } else {
                        cur = wstack_pop<SPILL>(stk_lds, stk_spill, stk_stride, --sp);
                    }
                } else if (d1 > 0.0f && d2 > 0.0f) {
                    if (d1 < d2) {  // left first
                        wstack_push<SPILL>(stk_lds, stk_spill, stk_stride, sp++, re);
                        cur = le;
                    } else {        // right first (ties included)
                        wstack_push<SPILL>(stk_lds, stk_spill, stk_stride, sp++, le);
                        cur = re;
                    }
                } else if (d1 > 0.0f) {
                    cur = le;
                } else if (d2 > 0.0f) {
                    cur = re;
                } else if (sp == 0) {
                    op = ENTRY ? OP_END : OP_H;
                    popped = true;
                } else {
                    cur = wstack_pop<SPILL>(stk_lds, stk_spill, stk_stride, --sp);
                }
                if (!popped && cur < 0) {
                    op = OP_LEAF, leaf_best = INF_;
                    if (STATS) st.leaves++;
                }
            }
        } else {
            // ---- cooperative leaf pass (hitArray, f5:243-256) ------------------------------------------------------
            // Every lane holding a leaf owns `rem` pending triangles.  The (owner, triangle) pairs of the whole wave are
            // laid out back to back (wave prefix sum), the first 64 become this pass's items and item k is tested by
            // lane k against its OWNER's ray (read from the LDS ray table).  Per owner the items reduce through one LDS
            // atomic min on (t bits << 32 | triangle index): the smallest t, the lowest index on a tie — what the
            // sequential scan with strict < keeps — and the owner then applies `t < best` (strict: an earlier leaf keeps
            // a tie, f5:278).  Triangles are consumed in index order, leaves in stack order: per ray nothing changes.
            const bool leaf = (op == OP_LEAF);
            const int ti = cur & 0x00ffffff;
            const int rem = leaf ? ((cur >> 24) & 0x7f) + 1 : 0;
            const int incl = wave_inclusive_sum(rem);
            const int start = incl - rem;
            int serve = 64 - start;
            serve = serve < 0 ? 0 : (serve > rem ? rem : serve);
            const int total = __shfl(incl, 63, 64);
            const int n_items = total < 64 ? total : 64;
            if (leaf) s_key[lane] = ~0ull;
            if (ENTRY && leaf) s_tie[lane] = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j)  // the reference builds leaves of up to 8 triangles (c5:843): no loop overhead for those
                if (j < serve) s_item[start + j] = ((ti + j) << 6) | lane;
            for (int j = 8; __any(j < serve); ++j)
                if (j < serve) s_item[start + j] = ((ti + j) << 6) | lane;
            __syncthreads();  // one wave per workgroup: a fence only
            int my_owner = 0, my_tri = 0;
            float my_t = -1.0f;  // < 0: this lane's item (if any) was not a hit
            if (lane < n_items) {
                const int it = s_item[lane];
                const int owner = it & 63, tri = it >> 6;
                const float4 r0 = s_ray[2 * owner], r1 = s_ray[2 * owner + 1];
                const float4* tb = (ENTRY && __float_as_int(r1.z) == 0) ? sc.ttri2 : sc.ttri;  // the owner's phase: which tree its leaf is of
                const float4 a = tb[tri * 3 + 0], b = tb[tri * 3 + 1], c = tb[tri * 3 + 2];
                float t;
                const bool hit = hitTriangleT(v3(a.x, a.y, a.z), v3(a.w, b.x, b.y), v3(b.z, b.w, c.x), v3(c.y, c.z, c.w),
                                              v3(r0.x, r0.y, r0.z), v3(r0.w, r1.x, r1.y), &t);
                if (hit) atomicMin(&s_key[owner], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)tri);
                if (STATS) s_t[lane] = hit ? t : INF_;
                if (ENTRY && hit) my_owner = owner, my_tri = tri, my_t = t;
            }
            __syncthreads();
            if (ENTRY) {  // two items of one owner at exactly its smallest distance: a tie the entry BVH's order must not decide
                if (my_t >= 0.0f) {
                    const unsigned long long k = s_key[my_owner];
                    if (__uint_as_float((unsigned)(k >> 32)) == my_t && (int)(unsigned)k != my_tri) s_tie[my_owner] = 1;
                }
                __syncthreads();
            }
            if (leaf && serve > 0) {
                bool done = false;
                if (STATS) {
                    st.tris += serve;
                    for (int j = 0; j < serve; ++j) {  // H = improvements inside hitArray, counted in scan order
                        const float tj = s_t[start + j];
                        if (tj < leaf_best) st.updates++, leaf_best = tj;
                    }
                }
                const unsigned long long key = s_key[lane];
                if (key != ~0ull) {
                    const float t = __uint_as_float((unsigned)(key >> 32));
                    if (ENTRY && phase == 0 && t == best && (int)(unsigned)key != best_tri) tie = true;
                    if (t < best) {
                        if (ENTRY) tie = (phase == 0) && (s_tie[lane] != 0);
                        best = t;
                        best_tri = (int)(unsigned)key;
                        // an env-visibility ray only needs isHit (f5:869); statistics builds count the full walk
                        if (CULL && KIND_ == K_SHADOW) done = true;
                    }
                }
                if (!done && serve < rem) {
                    cur = (int)(0x80000000u | ((unsigned)(rem - serve - 1) << 24) | (unsigned)(ti + serve));  // the rest of this leaf
                } else if (!done) {
                    if (sp == 0) {
                        done = true;
                    } else {
                        cur = wstack_pop<SPILL>(stk_lds, stk_spill, stk_stride, --sp);
                        if (cur >= 0) {
                            op = OP_INT;
                        } else {
                            leaf_best = INF_;
                            if (STATS) st.leaves++;
                        }
                    }
                }
                if (done) op = ENTRY ? OP_END : OP_H;
            }
            if (STATS && lane == 0) dg[10]++, dg[11] += n_items;
        }
        // ---- a walk ended: the next phase of the same ray, or the ray is done ----------------------------------------------
        if (ENTRY && op == OP_END) {
            if (ENTRY && phase == 0 && best_tri >= 0 && !(tie && KIND_ != K_SHADOW)) {
                // the entry BVH's closest (env-visibility rays: first) acceptable hit W: climb from its reference leaf
                const int w = sc.map2[best_tri];
                const unsigned lx = sc.leafx[w];
                // boxes to check explicitly: above them W's triangle lies inside every box by more than the margin, so the
                // reference's test cannot fail there ("short climb"); 255 = up to the root
                const int steps = (ceps <= sc.climb_eps) ? (int)(lx >> 24) : 255;
                best_tri = w;
                if (steps == 0) {
                    op = OP_H;
                } else {
                    phase = 1, cur = (int)(lx & 0x00ffffffu), sp = steps, op = OP_INT;
                }
            } else if (ENTRY && (phase == 0 ? best_tri >= 0 : phase == 2 && sp < 0)) {
                // (phase 0 with a tie, or a refuted check — marked by sp = -1 below —: the reference's own walk)
                phase = 2;
                best = INF_, best_tri = -1, sp = 0, cur = sc.root_entry;
                s_ray[2 * lane + 1].z = __int_as_float(phase);
                if (cur >= 0) {
                    op = OP_INT;
                } else {
                    op = OP_LEAF, leaf_best = INF_;
                    if (STATS) st.leaves++;
                }
            } else {
                op = OP_H;
            }
        }
#ifdef RT_OP_CYCLES
        oc[oc_pick > 2 ? 2 : oc_pick] += (unsigned long long)(clock64() - oc_t0);
#endif
        // ---- scheduler: ballot the populations, pick the next op ---------------------------------------------------
        const int nI = __popcll(__ballot(op == OP_INT)), nL = __popcll(__ballot(op == OP_LEAF));
        const int nH = __popcll(__ballot(op == OP_H));
        if ((nI | nL | nH) == 0) break;
        // traversal: internal steps until enough lanes hold a leaf to fill a cooperative leaf pass
        const int tb = (nL >= P.leaf_lanes || nI == 0) ? OP_LEAF : OP_INT, tn = nI + nL;
        // shade: when enough lanes wait for it, or when the lane-passes they have spent waiting are worth a shade pass
        // (long traversals: few lanes arrive per pass, and parking them until `refill` have gathered idles half the wave)
        waited += nH;
        pick = (tn == 0 || nH >= P.refill || waited >= P.wait_budget) ? OP_H : tb;
        if (pick == OP_H) waited = 0;
        if (STATS && lane == 0) dg[2 * pick]++, dg[2 * pick + 1] += (pick == OP_H) ? nH : ((pick == OP_INT) ? nI : nL);
        if (STATS) {  // lanes that have finished all their frames and idle until the wave ends
            const int nD = __popcll(__ballot(op == OP_DONE));
            if (lane == 0) dg[6]++, dg[7] += nD;
        }
    }
    // the longest-lived wave spans the whole launch: its lifetime in ticks / the launch's duration = the shader clock
    if (lane == 0) atomicMax(P.clk, (unsigned long long)(clock64() - wave_begin));
#ifdef RT_OP_CYCLES
    if (lane == 0) {
        oc[3] = (unsigned long long)(clock64() - oc_begin);
        for (int k = 0; k < 4; ++k) atomicAdd(&g_opcycles[k], oc[k]);
        atomicAdd(&g_opcycles[4], 1ull);
        atomicAdd(&g_opcycles[5], acq[0]), atomicAdd(&g_opcycles[6], acq[1]);
    }
#endif
#ifdef RT_SHADE_CYCLES
    if (lane == 0)
        for (int k = 0; k < 9; ++k) atomicAdd(&g_shadecycles[k], scy[k]);
#endif
    if (STATS) {
        flush_stats(P, st, paths);
        if (lane == 0)
            for (int k = 0; k < 12; ++k) atomicAdd(&P.stats[6 + k], (unsigned long long)dg[k]);
        if (lane == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(&P.stats[18 + k], (unsigned long long)sg[k]);
    }
}
#undef FIDX_
#undef BOUNCE_
#undef KIND_
#undef SET_KIND_

// sobolVec2(i, b) (f5:362-365) for the launch's frames (i = frameCounter + 1, f5:883) and the four bounces of chapter 5.
__global__ void __launch_bounds__(128) sobol_table_kernel(float2* out, int frame_begin, int frame_count) {
    const int t = blockIdx.x * 128 + threadIdx.x;
    if (t >= frame_count * 4) return;
    const unsigned frame = (unsigned)(frame_begin + t / 4);
    const unsigned gray = (frame + 1u) ^ ((frame + 1u) >> 1);
    out[t] = make_float2(sobol_dim(d_sobolV, 2 * (t & 3), gray), sobol_dim(d_sobolV, 2 * (t & 3) + 1, gray));
}

// CranleyPattersonRotation's per-pixel shift (f5:367-383) for every pixel of the image (two-kernel form: the BRDF-sampling step
// gathers it instead of carrying two registers through the whole path).
__global__ void __launch_bounds__(256) cp_table_kernel(float2* out, int width, int height) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)width * height) return;
    const int ix = (int)(p % (size_t)width), iy = (int)(p / (size_t)width);
    const float pixx = -1.0f + 2.0f * ((float)ix + 0.5f) / (float)width;
    const float pixy = -1.0f + 2.0f * ((float)iy + 0.5f) / (float)height;
    float u, v;
    cp_shift(pixx, pixy, width, height, &u, &v);
    out[p] = make_float2(u, v);
}

// ---- the first segment of every path: camera ray, closest hit, what follows up to the first secondary ray ------------------
// fshader.fsh main() up to its first hitBVH result (f5:926-951; f4:545-566; f3) and the start of bounce 0 (f5:856-866 / f4:513).
// One wave = one unit (an 8x8 tile x unit_frames frames), one lane = one pixel, frame after frame.  The 64 camera rays of a tile
// start at the eye and leave through neighbouring pixels: they walk the same nodes and test the same leaves, so the plain
// per-lane loop (hitBVH above: the reference's visit order, the closest-hit cull) runs with nearly every lane busy — the work
// the wave-scheduled path loop is worst at (it would spend an internal-node pass per node with about half its lanes idle).
//   miss  -> the sample is the environment colour (f5:945): stored, the path is over;
//   hit   -> the hit record (f5:205-219) and the first decision of bounce 0: chapter 5 draws the env sample (f5:862) — an
//            env-visibility ray if it faces the surface, else K_NONE (BRDF sampling next); chapters 3/4 draw the hemisphere
//            direction.  The state {hitPoint, next direction, N, V, seed, pixel, frame | kind, triangle} goes to the unit's
//            region of `states`, packed from its start (ballot + mbcnt); unit_count[unit] = how many.
// render_kernel<..., SEG> continues from those states.  Per sample the arithmetic and the RNG call order are unchanged.
// Launch shape: one wave per unit where the tree fits the LDS part of the stack (hardware dispatch balances the load best:
// persistent waves with a static stride were measured 25 % slower on config 5); deep trees (SPILL) bound the global spill buffer
// with persistent waves — as many as the path loop uses — that take their next unit from a counter.  ENTRY: hitBVH_entry.
template <int MODE, bool STATS, bool SPILL, bool ENTRY>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(ENTRY ? 5 : 8))) first_kernel(const RenderParams P, int n_units) {
    __shared__ int s_stack[kLdsDepth * kWave];
    const DevScene& sc = P.sc;
    const int lane = threadIdx.x;
    const int tilesx = (P.width + 7) >> 3, n_tiles = tilesx * ((P.height + 7) >> 3);
    LaneStack stk;
    stk.lds = &s_stack[lane], stk.lstride = kWave;
    stk.spill = SPILL ? (P.spill + ((long long)blockIdx.x * kWave + lane)) : nullptr, stk.stride = P.spill_stride;
    Stats st = {0, 0, 0, 0, 0};
    unsigned long long paths = 0;
    const size_t n_pixels = (size_t)P.width * P.height;
    const vec3 S = v3(P.eye[0], P.eye[1], P.eye[2]);
  const int n_groups = n_units / n_tiles;
  for (int work = blockIdx.x; work < n_units;) {
    // consecutive waves take the frame groups of ONE tile: nearly the same rays, so the nodes and leaves they fetch are the ones
    // their neighbours have just pulled into the caches (the unit's number — its region of `states` — stays group-major, as the
    // path loop's queues hand units out)
    const int tile = work / n_groups, group = work % n_groups;
    const int unit = group * n_tiles + tile;
    const int ix = (tile % tilesx) * 8 + (lane & 7), iy = (tile / tilesx) * 8 + (lane >> 3);
    const bool inside = ix < P.width && iy < P.height;
    const int f0 = group * P.unit_frames;
    const int nf = P.frame_count - f0 < P.unit_frames ? P.frame_count - f0 : P.unit_frames;
    float4* const region = P.states + (size_t)unit * (size_t)(P.unit_frames * kWave) * 4;
    const float pixx = -1.0f + 2.0f * ((float)ix + 0.5f) / (float)P.width;
    const float pixy = -1.0f + 2.0f * ((float)iy + 0.5f) / (float)P.height;
    int count = 0;  // wave-uniform: states written so far
    for (int k = 0; k < nf; ++k) {
        const int fidx = f0 + k;
        bool go_on = false;
        float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra, re = ra;
        if (inside) {
            unsigned seed = pixel_seed(pixx, pixy, P.width, P.height, (unsigned)(P.frame_begin + fidx));
            const float aax = (rand01(seed) - 0.5f) / (float)P.width;
            const float aay = (rand01(seed) - 0.5f) / (float)P.height;
            const float vx = pixx + aax, vy = pixy + aay, vz = -1.5f, vw = 0.0f;
            const float* m = P.cam;  // column-major: m[col*4+row]
            vec3 dir;
            dir.x = (m[0] * vx + m[4] * vy) + (m[8] * vz + m[12] * vw);
            dir.y = (m[1] * vx + m[5] * vy) + (m[9] * vz + m[13] * vw);
            dir.z = (m[2] * vx + m[6] * vy) + (m[10] * vz + m[14] * vw);
            const vec3 d = normalize(dir);
            float t;
            const int tri = ENTRY ? hitBVH_entry<STATS, SPILL>(sc, stk, S, d, &t, st)
                                  : hitBVH<false, TIE_SHADER, STATS, SPILL>(sc, stk, S, d, &t, st, P.stats_executed != 0);
            if (tri < 0) {
                vec3 color = v3(0);  // f5:945 / f4:557: the environment along the camera ray
                if (sc.hdr) {
                    float eu, ew;
                    toSphericalCoord(normalize(d), &eu, &ew);
                    const float4 c = sc.hdr[texel_index(ew, sc.hdrH) * sc.hdrW + texel_index(eu, sc.hdrW)];
                    color = v3(c.x, c.y, c.z);
                    if (MODE != RT_MODE_CHAP5) color = min3(color, v3(10));  // f4:162
                }
                float* o = P.samples + ((size_t)fidx * n_pixels + (size_t)iy * P.width + ix) * 3;
                o[0] = color.x, o[1] = color.y, o[2] = color.z;
                if (STATS) paths++;
            } else {
                const Hit nh = finish_hit<MODE>(sc, S, d, tri, t);
                const vec3 N = nh.normal;
                vec3 dn = v3(0);
                unsigned kind;
                if (MODE == RT_MODE_CHAP5) {
                    const float r1 = rand01(seed), r2 = rand01(seed);
                    const float4 ce = sc.cdir[texel_index(r2, sc.hdrH) * sc.hdrW + texel_index(r1, sc.hdrW)];
                    const vec3 Lh = v3(ce.x, ce.y, ce.z);
                    if (dot(N, Lh) > 0.0f)
                        dn = Lh, kind = K_SHADOW;
                    else
                        kind = K_NONE;
                } else {
                    dn = toNormalHemisphere(SampleHemisphereRand(seed), N), kind = K_BOUNCE;
                }
                go_on = true;
                ra = make_float4(nh.hitPoint.x, nh.hitPoint.y, nh.hitPoint.z, dn.x);
                rb = make_float4(dn.y, dn.z, N.x, N.y);
                rc = make_float4(N.z, -d.x, -d.y, -d.z);  // V = -viewDir
                re = make_float4(__uint_as_float(seed), __int_as_float(ix | (iy << 16)), __uint_as_float((unsigned)fidx | (kind << 30)), __int_as_float(tri));
            }
        }
        const unsigned long long m = __ballot(go_on);
        if (go_on) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            float4* o = region + (size_t)(count + rank) * 4;
            o[0] = ra, o[1] = rb, o[2] = rc, o[3] = re;
        }
        count += __popcll(m);
    }
    if (lane == 0) P.unit_count[unit] = count;
    if ((int)gridDim.x >= n_units) break;  // one wave per unit
    // persistent waves: units 0 .. gridDim.x - 1 went to the waves by number, the rest through a counter (a spare word of
    // the path loop's queue block, zeroed with it before the launch; the path loop starts after this kernel has ended)
    unsigned nx = 0;
    if (lane == 0) nx = __hip_atomic_fetch_add(P.queue + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    work = (int)gridDim.x + __builtin_amdgcn_readfirstlane((int)nx);
  }
    if (STATS) flush_stats(P, st, paths);
}

// The frame blend of f5:954-955 (or the plain sum of the spp-split multi-GPU mode): per pixel, over the launch's samples
// in frame order.
__global__ void __launch_bounds__(256) fold_kernel(const RenderParams P) {
    const size_t n_pixels = (size_t)P.width * P.height;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pixels) return;
    float* out = P.accum + p * 3;
    vec3 acc = v3(out[0], out[1], out[2]);
    for (int k = 0; k < P.frame_count; ++k) {
        const float* o = P.samples + ((size_t)k * n_pixels + p) * 3;
        const vec3 sample = v3(o[0], o[1], o[2]);
        const unsigned frame = (unsigned)(P.frame_begin + k);
        if (P.accumulate == RT_ACCUM_SUM)
            acc = acc + sample;
        else
            acc = mix3(acc, sample, 1.0f / (float)(int)(frame + 1u));
    }
    out[0] = acc.x, out[1] = acc.y, out[2] = acc.z;
}

// config 2: un-jittered camera rays, chapter-2 closest hit (c2:490-513)
template <bool STATS, bool SPILL>
__global__ void __launch_bounds__(kBlock) primary_kernel(const RenderParams P) {
    __shared__ int s_stack[kLdsDepth * kBlock];
    int px, py;
    const bool active = lane_pixel(P, &px, &py);
    LaneStack stk;
    stk.lds = &s_stack[threadIdx.x];
    stk.lstride = kBlock;
    stk.stride = P.spill_stride;
    stk.spill = SPILL ? (P.spill + ((long long)blockIdx.x * kBlock + threadIdx.x)) : nullptr;
    Stats st = {0, 0, 0, 0, 0};
    if (active) {
        const float vx = -1.0f + 2.0f * ((float)px + 0.5f) / (float)P.width;
        const float vy = -1.0f + 2.0f * ((float)py + 0.5f) / (float)P.height;
        const float vz = -1.5f, vw = 0.0f;
        const float* m = P.cam;
        vec3 dir;
        dir.x = (m[0] * vx + m[4] * vy) + (m[8] * vz + m[12] * vw);
        dir.y = (m[1] * vx + m[5] * vy) + (m[9] * vz + m[13] * vw);
        dir.z = (m[2] * vx + m[6] * vy) + (m[10] * vz + m[14] * vw);
        dir = normalize(dir);
        float t;
        const int tri = hitBVH<false, TIE_CHAP2, STATS, SPILL>(P.sc, stk, v3(P.eye[0], P.eye[1], P.eye[2]), dir, &t, st);
        const size_t o = (size_t)py * P.width + px;
        P.hit_tri[o] = tri;
        P.hit_dist[o] = t;
    }
    if (STATS) flush_stats(P, st, active ? 1 : 0);
}

// ---- host side ---------------------------------------------------------------------------------------------------

struct TimedLaunch {
    hipEvent_t a = nullptr, b = nullptr;  // around the path-loop kernel (render_kernel) / primary_kernel
    hipEvent_t a0 = nullptr;              // two-kernel form: before first_kernel (a0 .. a = its duration); else NULL
};
static thread_local std::vector<TimedLaunch> g_launches;
static void destroy_timed(TimedLaunch& tl) {
    (void)hipEventDestroy(tl.a), (void)hipEventDestroy(tl.b);
    if (tl.a0) (void)hipEventDestroy(tl.a0);
}
constexpr size_t kMaxTimedLaunches = 1024;  // bound on the events kept for rt_kernel_time_ms
static void remember_launch(const TimedLaunch& tl) {
    if (g_launches.size() >= kMaxTimedLaunches) {  // nobody is reading the timings: keep the most recent half
        for (size_t k = 0; k < kMaxTimedLaunches / 2; ++k) destroy_timed(g_launches[k]);
        g_launches.erase(g_launches.begin(), g_launches.begin() + kMaxTimedLaunches / 2);
    }
    g_launches.push_back(tl);
}
static thread_local int g_device = -1;
constexpr int kMaxDevices = 64;
// Which caller stream the overlapped launches of a device currently serve (see rt_render_device).
struct OverlapOwner {
    hipStream_t owner = nullptr;
    bool has_owner = false, disabled = false;
};
static std::mutex g_overlap_mutex;
static OverlapOwner g_overlap[kMaxDevices];
static int scratch_address(int device, unsigned char** out);
static int slot_streams(int device, hipStream_t out[2]);

// Makes the scene's device current for the duration of an entry point and restores the caller's afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == device) || (hipSetDevice(device) == hipSuccess);
        if (prev == device) prev = -1;  // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define GUARD_DEVICE(s_)                                                                                   \
    DeviceGuard guard_((s_)->device);                                                                      \
    if (!guard_.ok) return set_error(RT_ERR_HIP, "cannot make device %d current (%s:%d)", (s_)->device, __FILE__, __LINE__)

}  // namespace rta

using namespace rta;

struct rt_scene {
    int device = 0;
    int nT = 0, nN = 0;
    int max_stack = 0;  // stack entries the deepest traversal can hold (reference tree or 4-wide entry BVH)
    int ref_depth = 0;  // levels below the root of the reference tree
    DevScene dev{};
    void* d_ttri = nullptr;
    void* d_dmat = nullptr;
    void* d_tris = nullptr;
    void* d_nodes = nullptr;
    void* d_hdr = nullptr;
    void* d_cdir = nullptr;
    int* d_spill = nullptr;      // primary_kernel's global stack spill (deep trees)
    size_t spill_bytes = 0;
    // The path loop's workspace exists twice: consecutive launches alternate between the two slots, each with its own
    // internal stream, so that launch k+1 starts while launch k drains its last paths and its samples are folded.
    struct Slot {
        int* d_spill = nullptr;      // global stack spill (deep trees)
        size_t spill_bytes = 0;
        float* d_samples = nullptr;  // sample buffer (one launch: W x H x frames x RGB)
        size_t samples_bytes = 0;
        float2* d_sobol = nullptr;   // Sobol pairs of the launch's frames (kMaxLaunchFrames x 4)
        unsigned* d_queue = nullptr; // work-queue counters
        float4* d_states = nullptr;  // two-kernel form: path states per unit, the units' state counts, the per-pixel CP shifts
        size_t states_bytes = 0;
        int* d_count = nullptr;
        size_t count_bytes = 0;
        float2* d_cp = nullptr;
        size_t cp_bytes = 0;
        hipStream_t stream = nullptr;        // the path-loop kernel of this slot runs here
        hipEvent_t render_done = nullptr;    // recorded on `stream` after the path-loop kernel
        hipEvent_t fold_done = nullptr;      // recorded on the caller's stream after the fold that read this slot's samples
        bool fold_recorded = false;
    } slot[2];
    unsigned next_slot = 0;
    unsigned long long* d_clk = nullptr;  // longest wave lifetime (ticks) since the last rt_kernel_timing reset
    float* d_frame = nullptr;    // rt_render: device copy of the caller's host accumulation buffer
    size_t frame_bytes = 0;
    float* h_stage = nullptr;    // rt_render_multi: pinned host staging of this replica's sum buffer
    size_t stage_bytes = 0;
    int cu_count = 0;
    rt_tuning tuning{kLaunchFrames, kUnitFrames, kLeafLanes, kHandlerLanes, kWaitBudget, 0, kOverlapFrames, 0, 0};
    float cull_eps = 0.0f;       // scene part of the culling margin; +inf = the BVH / triangles do not allow culling
    void* d_nodes2 = nullptr;    // entry BVH (device-private, see host_fastbvh.cpp): records, triangles in its leaf order
    void* d_ttri2 = nullptr;
    void* d_map2 = nullptr;      // entry-BVH triangle -> caller's index; where each reference leaf / node box lives (the climb)
    void* d_leafx = nullptr;
    void* d_parentx = nullptr;
    bool has_entry = false;
    float cost_ref = 0.0f, cost_entry = 0.0f;  // expected internal-node visits of a random line: reference tree, entry BVH
    float root_box[6] = {0, 0, 0, 0, 0, 0};    // AA, BB of the caller's root node (valid only where the BVH's boxes were validated)
    bool root_box_ok = false;
    bool unit_frames_auto = true;              // rt_tuning.unit_frames was left at its default: the unit follows the view
    bool shade_lanes_auto = true;              // rt_tuning.shade_lanes was left at its default: it follows the scene and the mode
    bool launch_frames_auto = true;            // rt_tuning.launch_frames was left at its default: it follows the image size
    bool overlap_auto = true;                  // rt_tuning.overlap_frames was left at its default: launches overlap by their size in samples
};

extern "C" {

int rt_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_error(RT_ERR_NO_DEVICE, "rt_init: no HIP device available");
    if (device < 0 || device >= n) return set_error(RT_ERR_INVALID_ARG, "rt_init: device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    g_device = device;
    unsigned char* p = nullptr;
    return scratch_address(device, &p);  // this device's copy of the static scratch array (rt_render_device_noalloc)
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    DeviceGuard guard(s->device);
    (void)hipDeviceSynchronize();  // launches may still be in flight on the internal streams
    (void)hipFree(s->d_frame);
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    (void)hipFree(s->d_ttri), (void)hipFree(s->d_dmat), (void)hipFree(s->d_tris), (void)hipFree(s->d_nodes), (void)hipFree(s->d_hdr);
    (void)hipFree(s->d_cdir), (void)hipFree(s->d_spill), (void)hipFree(s->d_clk);
    (void)hipFree(s->d_nodes2), (void)hipFree(s->d_ttri2), (void)hipFree(s->d_map2), (void)hipFree(s->d_leafx), (void)hipFree(s->d_parentx);
    for (auto& sl : s->slot) {
        (void)hipFree(sl.d_spill), (void)hipFree(sl.d_samples), (void)hipFree(sl.d_queue), (void)hipFree(sl.d_sobol);
        (void)hipFree(sl.d_states), (void)hipFree(sl.d_count), (void)hipFree(sl.d_cp);
        if (sl.render_done) (void)hipEventDestroy(sl.render_done);
        if (sl.fold_done) (void)hipEventDestroy(sl.fold_done);
    }
    delete s;
}

int rt_scene_create(const float* tris, int nT, const float* nodes, int nN, const float* hdr_map, const float* hdr_cache,
                    int hdrW, int hdrH, rt_scene** out) {
    return rt_scene_create_ex(tris, nT, nodes, nN, hdr_map, hdr_cache, hdrW, hdrH, nullptr, out);
}

int rt_scene_create_ex(const float* tris, int nT, const float* nodes, int nN, const float* hdr_map, const float* hdr_cache,
                       int hdrW, int hdrH, const rt_scene_options* opt, rt_scene** out) {
    const int entry_mode = opt ? opt->entry_bvh : 0;
    if (!out) return set_error(RT_ERR_INVALID_ARG, "rt_scene_create: out is NULL");
    *out = nullptr;
    if (nT < 0 || nN < 0 || (nT > 0 && !tris) || (nN > 0 && !nodes))
        return set_error(RT_ERR_INVALID_ARG, "rt_scene_create: bad triangle/node arguments");
    if (nT >= (1 << 24)) return set_error(RT_ERR_INVALID_ARG, "rt_scene_create: %d triangles: fp32-encoded indices need < 2^24", nT);
    if ((hdr_map || hdr_cache) && (hdrW <= 0 || hdrH <= 0 || (long long)hdrW * hdrH > (1ll << 28)))
        return set_error(RT_ERR_INVALID_ARG, "rt_scene_create: bad HDR size %dx%d", hdrW, hdrH);
    if (hdr_cache && !hdr_map) return set_error(RT_ERR_INVALID_ARG, "rt_scene_create: hdr_cache without hdr_map");
    if (g_device < 0) {  // no rt_init on this thread: adopt the caller's current device
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) cur = 0;
        const int rc = rt_init(cur);
        if (rc != RT_OK) return rc;
    }

    // ---- everything derived from the caller's arrays, on the host (host_prep.cpp: all cores) ----------------------------
    ScenePrep P;
    {
        const int rc = prepare_scene(tris, nT, nodes, nN, hdr_map, hdr_cache, hdrW, hdrH, entry_mode, &P);
        if (rc != RT_OK) return rc;
    }
    const bool has_root = P.has_root, has_entry = P.has_entry;
    const int need_stack = P.need_stack, stack2 = P.stack2;
    const float cull_eps = P.cull_eps, cull_geo_max = P.cull_geo_max, climb_eps = P.climb_eps;
    const double cost_ref = P.cost_ref, cost_entry = P.cost_entry;
    const int root_entry2 = P.root_entry2;
    const RawBuf<float>&rec = P.rec, &ttri = P.ttri, &dmat = P.dmat, &rec2 = P.rec2, &ttri2 = P.ttri2;
    const std::vector<int>&map2 = P.map2, &parentx = P.parentx;
    const std::vector<unsigned>& leafx = P.leafx;

    rt_scene* s = new rt_scene();
    s->device = g_device;
    s->cull_eps = cull_eps;
    s->dev.cull_geo_max = cull_geo_max;
    s->nT = nT, s->nN = nN, s->max_stack = has_entry && stack2 > need_stack ? stack2 : need_stack;
    s->ref_depth = need_stack - 1;
    s->cost_ref = (float)cost_ref, s->cost_entry = (float)cost_entry, s->has_entry = has_entry;
    if (has_root && nN > 1 && cull_eps < __builtin_inff()) {  // (finite margin <=> every box holds its children and triangles, all finite)
        memcpy(s->root_box, nodes + 12 + 6, sizeof(s->root_box));  // BVHNode_encoded of node 1: childs, leafInfo, AA, BB
        s->root_box_ok = true;
    }
    auto fail = [&](int rc) {
        rt_scene_destroy(s);
        return rc;
    };
#define HIPCHK_S(expr)                                                                                               \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) return fail(set_error(RT_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__)); \
    } while (0)
    DeviceGuard guard(s->device);  // the caller's current device is restored on return
    if (!guard.ok) return fail(set_error(RT_ERR_HIP, "rt_scene_create: cannot make device %d current", s->device));
    HIPCHK_S(hipDeviceGetAttribute(&s->cu_count, hipDeviceAttributeMultiprocessorCount, s->device));
    HIPCHK_S(hipMalloc((void**)&s->d_clk, 64));
    HIPCHK_S(hipMemset(s->d_clk, 0, 64));
    // The two slots' streams are shared by all scenes on a device (slot_streams below): a process gets one low- and one
    // high-priority stream per device however many scenes it creates.
    hipStream_t shared[2] = {nullptr, nullptr};
    {
        const int rc = slot_streams(s->device, shared);
        if (rc != RT_OK) return fail(rc);
    }
    int slot_no = 0;
    for (auto& sl : s->slot) {
        HIPCHK_S(hipMalloc((void**)&sl.d_queue, 8 * 16 * sizeof(unsigned)));
        HIPCHK_S(hipMalloc((void**)&sl.d_sobol, (size_t)kMaxLaunchFrames * 4 * sizeof(float2)));
        sl.stream = shared[slot_no++];
        HIPCHK_S(hipEventCreateWithFlags(&sl.render_done, hipEventDisableTiming));
        HIPCHK_S(hipEventCreateWithFlags(&sl.fold_done, hipEventDisableTiming));
    }
    if (nT > 0) {
        HIPCHK_S(hipMalloc(&s->d_ttri, ttri.size() * 4));
        HIPCHK_S(hipMemcpy(s->d_ttri, ttri.data(), ttri.size() * 4, hipMemcpyHostToDevice));
        HIPCHK_S(hipMalloc(&s->d_dmat, dmat.size() * 4));
        HIPCHK_S(hipMemcpy(s->d_dmat, dmat.data(), dmat.size() * 4, hipMemcpyHostToDevice));
        HIPCHK_S(hipMalloc(&s->d_tris, (size_t)nT * 144));
        HIPCHK_S(hipMemcpy(s->d_tris, tris, (size_t)nT * 144, hipMemcpyHostToDevice));
    }
    if (!rec.empty()) {
        HIPCHK_S(hipMalloc(&s->d_nodes, rec.size() * 4));
        HIPCHK_S(hipMemcpy(s->d_nodes, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    }
    if (has_entry) {
        HIPCHK_S(hipMalloc(&s->d_nodes2, rec2.size() * 4));
        HIPCHK_S(hipMemcpy(s->d_nodes2, rec2.data(), rec2.size() * 4, hipMemcpyHostToDevice));
        HIPCHK_S(hipMalloc(&s->d_ttri2, ttri2.size() * 4));
        HIPCHK_S(hipMemcpy(s->d_ttri2, ttri2.data(), ttri2.size() * 4, hipMemcpyHostToDevice));
        HIPCHK_S(hipMalloc(&s->d_map2, map2.size() * 4));
        HIPCHK_S(hipMemcpy(s->d_map2, map2.data(), map2.size() * 4, hipMemcpyHostToDevice));
        HIPCHK_S(hipMalloc(&s->d_leafx, leafx.size() * 4));
        HIPCHK_S(hipMemcpy(s->d_leafx, leafx.data(), leafx.size() * 4, hipMemcpyHostToDevice));
        HIPCHK_S(hipMalloc(&s->d_parentx, parentx.size() * 4));
        HIPCHK_S(hipMemcpy(s->d_parentx, parentx.data(), parentx.size() * 4, hipMemcpyHostToDevice));
    }
    const size_t npx = (size_t)(hdr_map ? hdrW : 0) * (size_t)(hdr_map ? hdrH : 0);
    if (hdr_map) {
        HIPCHK_S(hipMalloc(&s->d_hdr, npx * 16));
        HIPCHK_S(hipMemcpy(s->d_hdr, P.hdr4.data(), npx * 16, hipMemcpyHostToDevice));
        if (hdr_cache) {
            HIPCHK_S(hipMalloc(&s->d_cdir, npx * 16));
            HIPCHK_S(hipMemcpy(s->d_cdir, P.cdir4.data(), npx * 16, hipMemcpyHostToDevice));
        }
    }
#undef HIPCHK_S
    s->dev.ttri = (const float4*)s->d_ttri;
    s->dev.dmat = (const float4*)s->d_dmat;
    s->dev.tris = (const float*)s->d_tris;
    s->dev.nodes = (const float4*)s->d_nodes;
    s->dev.hdr = (const float4*)s->d_hdr;
    s->dev.cdir = (const float4*)s->d_cdir;
    s->dev.hdrW = hdrW, s->dev.hdrH = hdrH;
    s->dev.has_root = has_root ? 1 : 0;
    s->dev.root_entry = P.root_entry;
    s->dev.cull_eps = cull_eps;
    s->dev.nodes2 = (const float4*)s->d_nodes2;
    s->dev.ttri2 = (const float4*)s->d_ttri2;
    s->dev.map2 = (const int*)s->d_map2, s->dev.leafx = (const unsigned*)s->d_leafx, s->dev.parentx = (const int*)s->d_parentx;
    s->dev.climb_eps = climb_eps;
    s->dev.root_entry2 = root_entry2;
    s->dev.has_entry = has_entry ? 1 : 0;
    (void)rt_scene_set_tuning(s, nullptr);  // the defaults for THIS scene (they depend on whether it has an entry BVH)
    *out = s;
    return RT_OK;
}

}  // extern "C"

namespace rta {

static int check_desc(const rt_scene* s, const rt_render_desc* d, const char* who) {
    if (!s || !d) return set_error(RT_ERR_INVALID_ARG, "%s: NULL scene or descriptor", who);
    // the path loop carries a pixel as (x | y << 16) in one int: x < 2^16, y < 2^15
    if (d->width <= 0 || d->height <= 0 || d->width > kMaxWidth || d->height > kMaxHeight)
        return set_error(RT_ERR_INVALID_ARG, "%s: bad image size %dx%d (1..%d x 1..%d)", who, d->width, d->height, kMaxWidth, kMaxHeight);
    return RT_OK;
}

// Device memory a launch needs beside the scene, the inputs and the outputs.  rt_render* takes it from buffers kept with
// the scene (allocated / grown on demand; two slots that consecutive launches alternate between, each with an internal
// stream); rt_render_device_noalloc carves one slot out of the static scratch array and stays on the caller's stream.
struct WorkSlot {
    unsigned* queue = nullptr;     // 8 work-queue counters, 64 B apart
    float2* sobol = nullptr;       // kMaxLaunchFrames x 4 Sobol pairs
    float* samples = nullptr;      // launch_frames x W x H x RGB
    int* spill = nullptr;          // global part of the traversal stacks (deep trees only)
    float4* states = nullptr;      // non-NULL: the two-kernel form (first_kernel -> render_kernel<..., SEG>) ...
    int* unit_count = nullptr;     // ... with its per-unit state counts
    float2* cp = nullptr;          // ... and the per-pixel Cranley-Patterson shifts
    rt_scene::Slot* owner = nullptr;  // the scene's slot these buffers belong to (NULL: the static scratch): its fold_done event
                                      // orders every reuse of the buffers, whichever stream the previous user ran on
    bool own_stream = false;       // run the path-loop kernel on owner->stream (overlapped launches), else on the caller's stream
};
struct Workspace {
    WorkSlot slot[2];
    int n_slots = 1;
    unsigned* next_slot = nullptr;  // which slot the next launch takes (kept with the scene)
    unsigned long long* clk = nullptr;  // longest wave lifetime (ticks)
    int launch_frames = 0;         // frames per kernel launch that `samples` holds
    long long spill_stride = 0;
    bool timed = false;            // bracket the path-loop kernel with HIP events (creates events: allocating callers only)
    bool camera_only = false;      // statistics of the camera rays alone: stop after first_kernel (two-kernel form only)
};

// Share of the image that the scene's root box can cover from this camera (1 when unknown).  Every camera ray outside the box's
// projection escapes at once (f5:945), so this is an upper bound on the share of paths that go on after their first ray.
// Camera model of f5:932-937: pixel (x, y) in [-1, 1]^2 looks along camera_rotate * (x, y, -1.5, 0) from `eye`.
static float screen_coverage(const rt_scene* s, const rt_render_desc* d) {
    if (!s->root_box_ok) return 1.0f;
    const float* m = d->camera_rotate;  // column-major
    double R[3][3];                     // R[r][c]: world component r of camera axis c
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) R[r][c] = m[c * 4 + r];
    for (int a = 0; a < 3; ++a)         // camera coordinates = R^T (p - eye) needs a rotation; anything else: no estimate
        for (int b = a; b < 3; ++b) {
            const double dp = R[0][a] * R[0][b] + R[1][a] * R[1][b] + R[2][a] * R[2][b];
            if (!(fabs(dp - (a == b ? 1.0 : 0.0)) < 1e-3)) return 1.0f;
        }
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (int k = 0; k < 8; ++k) {
        double q[3];
        for (int r = 0; r < 3; ++r) q[r] = (double)s->root_box[((k >> r) & 1) * 3 + r] - (double)d->eye[r];
        double v[3];
        for (int c = 0; c < 3; ++c) v[c] = R[0][c] * q[0] + R[1][c] * q[1] + R[2][c] * q[2];
        if (!(v[2] < -1e-9)) return 1.0f;  // a corner beside or behind the eye (or a NaN): the box may fill the view
        for (int c = 0; c < 2; ++c) {
            const double x = -1.5 * v[c] / v[2];
            lo[c] = x < lo[c] ? x : lo[c], hi[c] = x > hi[c] ? x : hi[c];
        }
    }
    double area = 1.0;
    const double pad[2] = {2.0 / (double)d->width, 2.0 / (double)d->height};  // the jitter moves a ray by up to half a pixel
    for (int c = 0; c < 2; ++c) {
        const double a = lo[c] - pad[c] < -1.0 ? -1.0 : lo[c] - pad[c], b = hi[c] + pad[c] > 1.0 ? 1.0 : hi[c] + pad[c];
        area *= b > a ? (b - a) * 0.5 : 0.0;
    }
    return (float)area;
}

// Frames per work unit.  rt_tuning.unit_frames when the caller set it; otherwise it follows the view: 2, 4 where the root box
// covers at most 40 % of the image, 8 at 15 % or less.  In a view that is mostly sky nearly every unit is empty and what costs is
// their number (and, in first_kernel, the per-unit set-up against two frames of work).  64-frame 1080p calls, config 4
// (tools/uf_probe.py, tools/coverage_probe.py): start-up camera, coverage 0.05, 97 % of the camera rays escape: 5.87 / 4.73 /
// 4.38 / 5.14 ms at 2 / 4 / 8 / 16 frames per unit; coverage 0.17 (92 % escape): 7.73 / 6.17 / 6.27; coverage 0.37 (86 %):
// 9.20 / 8.13 / 8.49.  Below ~70 % escaping rays the larger units lose (config 5, 25 % escape: -5 % at 8).  The box is an upper
// bound on what the rays can hit, so the choice errs towards 2.  Images do not depend on it (tested).
constexpr float kSkyCoverage8 = 0.15f, kSkyCoverage4 = 0.40f;
static int effective_unit_frames(const rt_scene* s, const rt_render_desc* d) {
    if (!s->unit_frames_auto) return s->tuning.unit_frames;
    const float c = screen_coverage(s, d);
    return c <= kSkyCoverage8 ? 8 : (c <= kSkyCoverage4 ? 4 : s->tuning.unit_frames);
}

// Frames per launch.  rt_tuning.launch_frames when the caller set it; otherwise 64 at 1080p and more on smaller images: a launch
// ramps up and drains for ~0.5 ms whatever its size, so what matters is its samples — 2^27 of them (1080p x 64) amortise it.
// 512-frame calls on config 5 (tools/lf_probe.py): 512 x 512 in launches of 64 frames 4842 Mpaths/s, of 512 frames 5838; 1280 x 720 in
// launches of 64 frames 5770, of 128 or more 5960; 1920 x 1080: 6030 from 32 to 512 frames per launch.  (The workspace bound of
// scene_workspace — samples + path states <= 12 GB — is the same 2^27 samples x 76 B; larger images get fewer frames there.)
static int effective_launch_frames(const rt_scene* s, const rt_render_desc* d) {
    if (!s->launch_frames_auto) return s->tuning.launch_frames;
    const long long px = (long long)d->width * d->height;
    long long lf = px > 0 ? ((long long)1 << 27) / px : kLaunchFrames;
    if (lf < kLaunchFrames) lf = kLaunchFrames;
    if (lf > kMaxLaunchFrames) lf = kMaxLaunchFrames;
    return (int)lf;
}

// Does a launch of `lf` frames overlap its neighbours (two slots, internal streams)?  rt_tuning.overlap_frames when the caller set
// it (in frames); otherwise by size: below 2^26 samples — 1080p x 32 — the ramp-up and drain that overlapping hides are worth more
// than the two path loops cost each other.  Config 5, overlap off / on (tools/ov_probe.py, tools/fpc_probe.py): 1080p x 32 frames
// 5860 / 6030 Mpaths/s, x 64 6040 / 5930; 1280 x 720 x 64 (59 M samples) 5779 / 5954, x 128 5986 / 5895; 512 x 512 x 64 4853 / 5545,
// x 128 5386 / 5701, x 256 (2^26) 5700 / 5756.
// A scene with an entry BVH gains far more (its coherent camera-ray kernel runs in what the neighbour's latency-bound path loop
// leaves idle: the 10 M-triangle scene, 4K x 8 frames, 895 -> 1057 Mpaths/s; x 16 frames 1190 overlapped): every launch its workspace
// bound allows overlaps there.
constexpr long long kOverlapMaxSamples = (long long)1 << 26, kOverlapMaxSamplesBig = (long long)1 << 28;
static bool launch_overlaps(const rt_scene* s, const rt_render_desc* d, int lf) {
    if (!s->overlap_auto) return lf < s->tuning.overlap_frames;
    return (long long)d->width * d->height * lf < (s->has_entry ? kOverlapMaxSamplesBig : kOverlapMaxSamples);
}

static void fill_params(const rt_scene* s, const rt_render_desc* d, RenderParams* P) {
    memset(P, 0, sizeof(*P));
    P->sc = s->dev;
    P->width = d->width, P->height = d->height;
    P->accumulate = d->accumulate;
    memcpy(P->eye, d->eye, sizeof(P->eye));
    memcpy(P->cam, d->camera_rotate, sizeof(P->cam));
    P->leaf_lanes = s->tuning.leaf_lanes;
    P->wait_budget = s->tuning.wait_budget;
    // (chapter 3's diffuse shade pass is a fraction of chapter 5's: it pays earlier — config-5 scene, 1080p: 7766 -> 8040 Mpaths/s at 36
    // waiting lanes instead of 48; chapter 4: 48 or 40 within 1.5 % either way, scene-dependent; tools/mode_probe.py)
    P->refill = (s->shade_lanes_auto && d->mode == RT_MODE_CHAP3 && !s->has_entry) ? kHandlerLanesChap3 : s->tuning.shade_lanes;
    P->unit_frames = effective_unit_frames(s, d);
}

// render_kernel: persistent single-wave workgroups, as many as the device can hold (32 wave slots per CU is the hardware
// ceiling; workgroups beyond the kernel's real occupancy start late, find the queue empty and leave)
static int render_blocks(const rt_scene* s) { return s->cu_count * 32; }
// bytes of global stack spill `threads` lanes need on this scene (0: the tree fits the LDS part of the stack)
static size_t spill_bytes_needed(const rt_scene* s, size_t threads) {
    return s->max_stack > kLdsDepth ? threads * (size_t)(s->max_stack - kLdsDepth) * sizeof(int) : 0;
}
static int grow_scene_spill(rt_scene* s, size_t threads, int** spill, long long* stride) {
    const size_t bytes = spill_bytes_needed(s, threads);
    *spill = nullptr, *stride = 0;
    if (bytes == 0) return RT_OK;
    if (bytes > s->spill_bytes) {
        (void)hipFree(s->d_spill);
        s->d_spill = nullptr, s->spill_bytes = 0;
        HIPCHK(hipMalloc((void**)&s->d_spill, bytes));
        s->spill_bytes = bytes;
    }
    *spill = s->d_spill, *stride = (long long)threads;
    return RT_OK;
}

static int check_render_args(const rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, const float* accum_dev) {
    int rc = check_desc(s, d, "rt_render");
    if (rc != RT_OK) return rc;
    if (!accum_dev) return set_error(RT_ERR_INVALID_ARG, "rt_render: accum is NULL");
    if (frame_begin < 0 || frame_count < 0) return set_error(RT_ERR_INVALID_ARG, "rt_render: negative frame range");
    if (frame_count > 0x7fffffff - frame_begin) return set_error(RT_ERR_INVALID_ARG, "rt_render: frames %d + %d run past 2^31 - 1", frame_begin, frame_count);
    if (d->mode != RT_MODE_CHAP3 && d->mode != RT_MODE_CHAP4 && d->mode != RT_MODE_CHAP5)
        return set_error(RT_ERR_INVALID_ARG, "rt_render: unknown mode %d", d->mode);
    if (d->accumulate != RT_ACCUM_RUNNING_MEAN && d->accumulate != RT_ACCUM_SUM)
        return set_error(RT_ERR_INVALID_ARG, "rt_render: unknown accumulate mode %d", d->accumulate);
    if (d->mode == RT_MODE_CHAP5 && !s->d_cdir)
        return set_error(RT_ERR_INVALID_ARG, "rt_render: mode 5 needs hdr_map and hdr_cache in the scene");
    const int mb = d->max_bounce > 0 ? d->max_bounce : (d->mode == RT_MODE_CHAP5 ? 4 : 6);
    if (d->mode == RT_MODE_CHAP5 && mb > 4)  // sobolVec2(i, bounce) reads V[(2*bounce+1)*32 + j] of the 8-dimension table (f5:339, 883)
        return set_error(RT_ERR_INVALID_ARG, "rt_render: mode 5 supports max_bounce <= 4 (the reference's Sobol table has 8 dimensions), got %d", mb);
    return RT_OK;
}

// The launches of one rt_render* call on a prepared workspace (arguments already checked, scene's device current).
// Allocates nothing unless ws.timed (HIP events).
template <bool STATS>
static int launch_render(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, float* accum_dev,
                         unsigned long long* stats_dev, hipStream_t stream, const Workspace& ws, int stats_executed = 0) {
    RenderParams P;
    fill_params(s, d, &P);
    P.stats_executed = stats_executed;
    P.max_bounce = d->max_bounce > 0 ? d->max_bounce : (d->mode == RT_MODE_CHAP5 ? 4 : 6);
    P.accum = accum_dev;
    P.stats = stats_dev;
    P.clk = ws.clk;
    P.spill_stride = ws.spill_stride;
    const int blocks = render_blocks(s);
    const int launch_frames = ws.launch_frames;
    const bool spill = ws.slot[0].spill != nullptr;
    // the instantiation that knows the entry BVH: not for the reference-walk statistics (they never leave the reference tree)
    const bool entry = P.sc.has_entry != 0 && (!STATS || stats_executed != 0);
    const int n_tiles = ((d->width + 7) / 8) * ((d->height + 7) / 8);
    const int fold_blocks = (int)(((size_t)d->width * d->height + 255) / 256);
    for (int done = 0; done < frame_count; done += launch_frames) {
        P.frame_begin = frame_begin + done;
        P.frame_count = frame_count - done < launch_frames ? frame_count - done : launch_frames;
        const long long n_units = (long long)n_tiles * ((P.frame_count + P.unit_frames - 1) / P.unit_frames);
        const int grid = (int)(n_units < blocks ? n_units : blocks);  // persistent waves: at most what the device holds
        // this launch's slot; with two slots the path-loop kernel runs on the slot's own stream: it needs nothing from the
        // caller's stream (the scene is immutable, `accum` is only touched by the fold), so it may start while the previous
        // launch — other slot — is still draining its last paths; the fold then waits for it on the caller's stream
        const WorkSlot& w = ws.slot[ws.next_slot ? ((*ws.next_slot)++ % (unsigned)ws.n_slots) : 0];
        P.queue = w.queue, P.samples = w.samples, P.sobol = w.sobol, P.spill = w.spill;
        P.states = w.states, P.unit_count = w.unit_count, P.cp = w.cp;
        // two-kernel form: where scene_workspace gave the slot a path-state buffer (launches of two or more frames, unless rt_tuning.single_kernel)
        const bool seg = w.states != nullptr;
        if (ws.camera_only && !seg)
            return set_error(RT_ERR_INVALID_ARG, "rt_render_stats_camera: this scene renders with the one-kernel path loop (rt_tuning.single_kernel = 1): the camera rays are not a kernel of their own");
        const bool piped = w.owner && w.own_stream;
        const hipStream_t rstream = piped ? w.owner->stream : stream;
        // the slot's previous user — overlapped or not, on whatever stream — has folded its samples (a wait on an event
        // recorded on the same stream costs nothing)
        if (w.owner && w.owner->fold_recorded) HIPCHK(hipStreamWaitEvent(rstream, w.owner->fold_done, 0));
        HIPCHK(hipMemsetAsync(w.queue, 0, 8 * 16 * sizeof(unsigned), rstream));
        if (d->mode == RT_MODE_CHAP5)
            hipLaunchKernelGGL(sobol_table_kernel, dim3((P.frame_count * 4 + 127) / 128), dim3(128), 0, rstream, w.sobol, P.frame_begin, P.frame_count);
        TimedLaunch tl{};
        const bool timed = ws.timed && !STATS;
        if (seg) {
            if (d->mode == RT_MODE_CHAP5)
                hipLaunchKernelGGL(cp_table_kernel, dim3(fold_blocks), dim3(256), 0, rstream, w.cp, d->width, d->height);
            if (timed) {
                HIPCHK(hipEventCreate(&tl.a0));
                HIPCHK(hipEventRecord(tl.a0, rstream));
            }
            // a wave per unit (tile x unit_frames frames): it leaves unit_count[unit] and the unit's states for the path loop.
            // Deep trees: persistent waves (the global part of the stacks is sized for the path loop's thread count)
            const unsigned fgrid = (unsigned)(spill ? (n_units < blocks ? n_units : blocks) : n_units);
#define FIRST3(MODE_, SPILL_, ENTRY_) hipLaunchKernelGGL((first_kernel<MODE_, STATS, SPILL_, ENTRY_>), dim3(fgrid), dim3(kWave), 0, rstream, P, (int)n_units)
#define FIRST2(MODE_, SPILL_) do { if (entry) FIRST3(MODE_, SPILL_, true); else FIRST3(MODE_, SPILL_, false); } while (0)
#define FIRST(MODE_) do { if (spill) FIRST2(MODE_, true); else FIRST2(MODE_, false); } while (0)
            if (d->mode == RT_MODE_CHAP5)
                FIRST(RT_MODE_CHAP5);
            else if (d->mode == RT_MODE_CHAP4)
                FIRST(RT_MODE_CHAP4);
            else
                FIRST(RT_MODE_CHAP3);
#undef FIRST
#undef FIRST2
#undef FIRST3
            HIPCHK(hipGetLastError());
        }
        if (ws.camera_only) continue;
        if (timed) {
            HIPCHK(hipEventCreate(&tl.a));
            HIPCHK(hipEventCreate(&tl.b));
            HIPCHK(hipEventRecord(tl.a, rstream));
        }
#define LAUNCH3(MODE_, SPILL_, ENTRY_)                                                                          \
    do {                                                                                                        \
        if (seg)                                                                                                \
            hipLaunchKernelGGL((render_kernel<MODE_, STATS, SPILL_, ENTRY_, true>), dim3(grid), dim3(kWave), 0, rstream, P);  \
        else                                                                                                    \
            hipLaunchKernelGGL((render_kernel<MODE_, STATS, SPILL_, ENTRY_, false>), dim3(grid), dim3(kWave), 0, rstream, P); \
    } while (0)
#define LAUNCH2(MODE_, SPILL_)                 \
    do {                                       \
        if (entry)                             \
            LAUNCH3(MODE_, SPILL_, true);      \
        else                                   \
            LAUNCH3(MODE_, SPILL_, false);     \
    } while (0)
#define LAUNCH(MODE_)                  \
    do {                               \
        if (spill)                     \
            LAUNCH2(MODE_, true);      \
        else                           \
            LAUNCH2(MODE_, false);     \
    } while (0)
        if (d->mode == RT_MODE_CHAP5)
            LAUNCH(RT_MODE_CHAP5);
        else if (d->mode == RT_MODE_CHAP4)
            LAUNCH(RT_MODE_CHAP4);
        else
            LAUNCH(RT_MODE_CHAP3);
#undef LAUNCH
#undef LAUNCH2
#undef LAUNCH3
        HIPCHK(hipGetLastError());
        if (timed) {
            HIPCHK(hipEventRecord(tl.b, rstream));
            remember_launch(tl);
        }
        if (piped) {
            HIPCHK(hipEventRecord(w.owner->render_done, rstream));
            HIPCHK(hipStreamWaitEvent(stream, w.owner->render_done, 0));
        }
        hipLaunchKernelGGL(fold_kernel, dim3(fold_blocks), dim3(256), 0, stream, P);
        HIPCHK(hipGetLastError());
        if (w.owner) {
            HIPCHK(hipEventRecord(w.owner->fold_done, stream));
            w.owner->fold_recorded = true;
        }
    }
    return RT_OK;
}

// Before a slot's buffer is freed: everything that touches it has finished (the fold is the last user of a launch's
// buffers and waits for its path-loop kernel).
static void quiesce_slot(rt_scene::Slot& sl) {
    if (!sl.fold_recorded) return;  // never used since the last time
    (void)hipEventSynchronize(sl.fold_done);
    sl.fold_recorded = false;
}

// Workspace of the allocating entry points: buffers kept with the scene, grown when a call needs more.  `pipelined`:
// both slots, path-loop kernels on the slots' streams; otherwise slot 0 alone on the caller's stream (statistics passes).
static int scene_workspace(rt_scene* s, const rt_render_desc* d, int frame_count, Workspace* ws, bool pipelined, bool camera_only = false) {
    // frames per kernel launch: bounded by the sample buffer (W x H x frames x RGB fp32)
    int launch_frames = effective_launch_frames(s, d);
    if (launch_frames > frame_count) launch_frames = frame_count;
    const size_t per_frame = (size_t)d->width * d->height * 3 * sizeof(float);
    const size_t threads = (size_t)render_blocks(s) * kWave;
    const size_t spill = spill_bytes_needed(s, threads);
    // two-kernel form (first_kernel -> render_kernel<..., SEG>): every sample slot of a unit may leave a 64-byte path state
    // — except for small launches, the reference's own calling pattern above all (one frame per display(), c5:706-760, in a 512 x 512
    // window): a launch of a few million samples is all ramp-up and drain, and a second kernel in front of it only adds to both.
    // Config 5, ms per call in two kernels / in one: 1080p x 1 frame 0.659 / 0.605, x 2 1.02 / 1.02, x 4 1.70 / 1.88; 720p x 4
    // 0.98 / 0.92, x 8 1.56 / 1.70; 512 x 512 x 8 0.73 / 0.64, x 64 3.47 / 4.18 (tools/fpc_probe.py, tools/small_probe.py): the
    // crossover sits at about four million samples per launch whatever their shape.
    const bool seg = s->tuning.single_kernel != 1 &&
                     (s->tuning.single_kernel == 2 || (size_t)d->width * d->height * (size_t)launch_frames > kTwoKernelMinSamples || camera_only);
    const size_t n_tiles = (size_t)((d->width + 7) / 8) * (size_t)((d->height + 7) / 8);
    const size_t state_per_frame = seg ? n_tiles * kWave * 4 * sizeof(float4) : 0;
    while (launch_frames > 1 && (per_frame + state_per_frame) * launch_frames > kMaxSampleBytes) launch_frames >>= 1;
    const int uf = effective_unit_frames(s, d);
    const size_t n_units = n_tiles * (size_t)((launch_frames + uf - 1) / uf);
    const size_t states_bytes = seg ? n_units * (size_t)uf * kWave * 4 * sizeof(float4) : 0;
    const size_t count_bytes = seg ? n_units * sizeof(int) : 0;
    const size_t cp_bytes = seg ? (size_t)d->width * d->height * sizeof(float2) : 0;
    // short launches are dominated by ramp-up and drain (a launch ends with every wave finishing its last paths at falling
    // lane utilisation): those overlap their neighbours.  Long launches stay on the caller's stream, one at a time, so that
    // a kernel's duration means what it says (rt_kernel_time_ms, rocprofv3).
    pipelined = pipelined && launch_overlaps(s, d, launch_frames);
    ws->n_slots = pipelined ? 2 : 1;
    ws->next_slot = pipelined ? &s->next_slot : nullptr;
    for (int k = 0; k < ws->n_slots; ++k) {
        rt_scene::Slot& sl = s->slot[k];
        if (per_frame * launch_frames > sl.samples_bytes) {
            quiesce_slot(sl);
            (void)hipFree(sl.d_samples);
            sl.d_samples = nullptr, sl.samples_bytes = 0;
            if (hipMalloc((void**)&sl.d_samples, per_frame * launch_frames) != hipSuccess)
                return set_error(RT_ERR_OOM, "rt_render: cannot allocate the %zu-byte sample buffer", per_frame * launch_frames);
            sl.samples_bytes = per_frame * launch_frames;
        }
        if (spill > sl.spill_bytes) {
            quiesce_slot(sl);
            (void)hipFree(sl.d_spill);
            sl.d_spill = nullptr, sl.spill_bytes = 0;
            HIPCHK(hipMalloc((void**)&sl.d_spill, spill));
            sl.spill_bytes = spill;
        }
        if (states_bytes > sl.states_bytes || count_bytes > sl.count_bytes || cp_bytes > sl.cp_bytes) {
            quiesce_slot(sl);
            if (states_bytes > sl.states_bytes) {
                (void)hipFree(sl.d_states);
                sl.d_states = nullptr, sl.states_bytes = 0;
                if (hipMalloc((void**)&sl.d_states, states_bytes) != hipSuccess)
                    return set_error(RT_ERR_OOM, "rt_render: cannot allocate the %zu-byte path-state buffer (rt_tuning.single_kernel = 1 renders without it)", states_bytes);
                sl.states_bytes = states_bytes;
            }
            if (count_bytes > sl.count_bytes) {
                (void)hipFree(sl.d_count);
                sl.d_count = nullptr, sl.count_bytes = 0;
                HIPCHK(hipMalloc((void**)&sl.d_count, count_bytes));
                sl.count_bytes = count_bytes;
            }
            if (cp_bytes > sl.cp_bytes) {
                (void)hipFree(sl.d_cp);
                sl.d_cp = nullptr, sl.cp_bytes = 0;
                HIPCHK(hipMalloc((void**)&sl.d_cp, cp_bytes));
                sl.cp_bytes = cp_bytes;
            }
        }
        ws->slot[k].states = seg ? sl.d_states : nullptr, ws->slot[k].unit_count = sl.d_count, ws->slot[k].cp = sl.d_cp;
        ws->slot[k].queue = sl.d_queue, ws->slot[k].sobol = sl.d_sobol, ws->slot[k].samples = sl.d_samples;
        ws->slot[k].spill = spill ? sl.d_spill : nullptr;
        ws->slot[k].owner = &sl, ws->slot[k].own_stream = pipelined;
    }
    ws->clk = s->d_clk;
    ws->spill_stride = spill ? (long long)threads : 0;
    ws->launch_frames = launch_frames;
    ws->timed = true;
    return RT_OK;
}

// ---- the static scratch array of rt_render_device_noalloc ---------------------------------------------------------------
// One array per GPU (each device that loads this code object holds its own copy at its own address, until the process
// ends).  Host code never uses the symbol's name as an address: the per-device address comes from hipGetSymbolAddress,
// once per device, and is what kernels, memsets and copies receive.
constexpr size_t kScratchBytes = (size_t)64 << 20;
constexpr size_t kScratchQueueOff = 0;                                   // 8 counters, 64 B apart (512 B)
constexpr size_t kScratchClkOff = 1024;                                  // longest wave lifetime (8 B)
constexpr size_t kScratchSobolOff = 4096;                                // kMaxLaunchFrames x 4 x float2 = 32 KiB
constexpr size_t kScratchBodyOff = 65536;                                // stack spill (deep trees), then the sample buffer
static_assert(kScratchSobolOff + (size_t)kMaxLaunchFrames * 4 * sizeof(float2) <= kScratchBodyOff, "scratch header layout");
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

static std::mutex g_scratch_mutex;
static unsigned char* g_scratch_ptr[kMaxDevices];  // per device: the address of ITS copy of `scratch`
// In-use guard of the scratch array: the last rt_render_device_noalloc call on a device records g_scratch_done on its stream
// after its last fold.  A later call on the SAME stream is ordered behind it by the stream; a call on another stream while
// that event has not completed would race on the array and is refused.  The event is created with the address look-up
// (rt_init / rt_scene_create), so the entry point itself creates nothing.
static hipEvent_t g_scratch_done[kMaxDevices];
static hipStream_t g_scratch_stream[kMaxDevices];
static bool g_scratch_used[kMaxDevices];

// The two internal streams overlapped launches run on (one per workspace slot), per device, created on first use and kept
// until the process ends; every scene on the device uses them.  They have different priorities: the runtime keeps separate
// hardware-queue pools per priority, so the two can never be mapped onto one hardware queue (GPU_MAX_HW_QUEUES is 4 per
// process and streams share queues; two equal-priority streams that land on the same queue would run their kernels one after
// the other — observed).  They are shared rather than per scene because every priority stream may bring a hardware queue of
// its own: with ten of them (five scenes) the device's queues were oversubscribed and a chain of one-frame launches — kernel on
// a slot stream, fold on the caller's stream, each waiting for the other's event — stopped making progress (tools/soak.py
// found it).  Sharing only orders different scenes' overlapped launches among themselves.  `device` must be current.
static std::mutex g_streams_mutex;
static hipStream_t g_slot_stream[64][2];
static int slot_streams(int device, hipStream_t out[2]) {
    if (device < 0 || device >= 64) return set_error(RT_ERR_INVALID_ARG, "device %d: the stream table holds 64 devices", device);
    std::lock_guard<std::mutex> lock(g_streams_mutex);
    if (!g_slot_stream[device][0]) {
        int prio_least = 0, prio_greatest = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        hipStream_t a = nullptr, b = nullptr;
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&a, hipStreamNonBlocking, prio_least);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&b, hipStreamNonBlocking, prio_greatest);
        if (e != hipSuccess) {
            if (a) (void)hipStreamDestroy(a);
            return set_error(RT_ERR_HIP, "creating the internal streams on device %d: %s", device, hipGetErrorString(e));
        }
        g_slot_stream[device][0] = a, g_slot_stream[device][1] = b;
    }
    out[0] = g_slot_stream[device][0], out[1] = g_slot_stream[device][1];
    return RT_OK;
}

// `device` must be current.
static int scratch_address(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return set_error(RT_ERR_INVALID_ARG, "device %d: the scratch table holds %d devices", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return set_error(RT_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
        hipEvent_t ev = nullptr;
        const hipError_t e2 = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e2 != hipSuccess) return set_error(RT_ERR_HIP, "creating the scratch guard event on device %d: %s", device, hipGetErrorString(e2));
        g_scratch_done[device] = ev;
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

}  // namespace rta

extern "C" {

int rt_scene_get_tuning(const rt_scene* s, rt_tuning* out) {
    if (!s || !out) return set_error(RT_ERR_INVALID_ARG, "rt_scene_get_tuning: NULL argument");
    *out = s->tuning;
    if (s->unit_frames_auto) out->unit_frames = 0;  // (0 = chosen per view: a get / modify / set round trip keeps it that way)
    if (s->shade_lanes_auto) out->shade_lanes = 0;  // (0 = chosen per scene and mode)
    if (s->launch_frames_auto) out->launch_frames = 0;  // (0 = chosen per image size)
    if (s->overlap_auto) out->overlap_frames = 0;       // (0 = by the launch's size in samples)
    return RT_OK;
}

int rt_scene_set_tuning(rt_scene* s, const rt_tuning* t) {
    if (!s) return set_error(RT_ERR_INVALID_ARG, "rt_scene_set_tuning: NULL scene");
    // Scheduler thresholds: 8 / 48 is the optimum where the scene sits in the caches (config 5, re-swept each round).  A scene
    // that needed an entry BVH is far beyond them: its walks are chains of cache misses, few lanes come back per pass, and
    // parking them until 48 have gathered idles too much of the wave — 16 / 24 there: the 10 M-triangle scene at 4K, 8 frames
    // per step, 999 -> 1117 and 1062 -> 1185 Mpaths/s (the two slot parities; plateau over 12..16 / 16..24, tools/sweep_tuning.py --syn)
    const rt_tuning def = {kLaunchFrames, kUnitFrames, s->has_entry ? kLeafLanesBig : kLeafLanes, s->has_entry ? kHandlerLanesBig : kHandlerLanes,
                           kWaitBudget, 0, kOverlapFrames, 0, 0};
    rt_tuning n = t ? *t : def;
    if (n.launch_frames <= 0) n.launch_frames = def.launch_frames;
    if (n.unit_frames <= 0) n.unit_frames = def.unit_frames;
    if (n.leaf_lanes <= 0) n.leaf_lanes = def.leaf_lanes;
    if (n.shade_lanes <= 0) n.shade_lanes = def.shade_lanes;
    if (n.wait_budget <= 0) n.wait_budget = def.wait_budget;
    if (n.launch_frames > kMaxLaunchFrames) n.launch_frames = kMaxLaunchFrames;
    if (n.unit_frames > kMaxLaunchFrames) n.unit_frames = kMaxLaunchFrames;
    if (n.leaf_lanes > kWave) n.leaf_lanes = kWave;
    if (n.shade_lanes > kWave) n.shade_lanes = kWave;
    n.no_cull = n.no_cull > 0 ? 1 : 0;
    if (n.overlap_frames <= 0) n.overlap_frames = def.overlap_frames;
    s->tuning = n;
    s->dev.cull_eps = n.no_cull ? __builtin_inff() : s->cull_eps;
    n.no_entry_bvh = n.no_entry_bvh > 0 ? 1 : 0;
    n.single_kernel = n.single_kernel >= 2 ? 2 : (n.single_kernel == 1 ? 1 : 0);  // 0: by launch size, 1: one kernel, 2: two kernels
    s->tuning = n;
    s->unit_frames_auto = !(t && t->unit_frames > 0);
    s->shade_lanes_auto = !(t && t->shade_lanes > 0);
    s->launch_frames_auto = !(t && t->launch_frames > 0);
    s->overlap_auto = !(t && t->overlap_frames > 0);
    s->dev.has_entry = (s->has_entry && !n.no_entry_bvh && !n.no_cull) ? 1 : 0;
    return RT_OK;
}

int rt_scene_get_info(const rt_scene* s, rt_scene_info* out) {
    if (!s || !out) return set_error(RT_ERR_INVALID_ARG, "rt_scene_get_info: NULL argument");
    out->n_triangles = s->nT, out->n_nodes = s->nN;
    out->bvh_depth = s->ref_depth;
    out->stack_spill = s->max_stack > kLdsDepth ? 1 : 0;
    out->cull_enabled = (s->dev.cull_eps < __builtin_inff()) ? 1 : 0;
    out->cull_eps = s->cull_eps;
    out->device = s->device;
    out->entry_bvh = s->dev.has_entry;
    out->cost_reference_tree = s->cost_ref, out->cost_entry_bvh = s->cost_entry;
    return RT_OK;
}

int rt_scene_screen_coverage(const rt_scene* s, const rt_render_desc* d, float* coverage, int* unit_frames) {
    if (!s || !d || !coverage) return set_error(RT_ERR_INVALID_ARG, "rt_scene_screen_coverage: NULL argument");
    if (d->width <= 0 || d->height <= 0) return set_error(RT_ERR_INVALID_ARG, "rt_scene_screen_coverage: bad image size %dx%d", d->width, d->height);
    *coverage = screen_coverage(s, d);
    if (unit_frames) *unit_frames = effective_unit_frames(s, d);
    return RT_OK;
}

int rt_scene_launch_shape(const rt_scene* s, const rt_render_desc* d, int frame_count, rt_launch_shape* out) {
    if (!s || !d || !out) return set_error(RT_ERR_INVALID_ARG, "rt_scene_launch_shape: NULL argument");
    if (d->width <= 0 || d->height <= 0 || frame_count <= 0) return set_error(RT_ERR_INVALID_ARG, "rt_scene_launch_shape: bad image size %dx%d or frame count %d", d->width, d->height, frame_count);
    int lf = effective_launch_frames(s, d);
    if (lf > frame_count) lf = frame_count;
    const size_t per_frame = (size_t)d->width * d->height * 3 * sizeof(float);
    const bool want_seg = s->tuning.single_kernel != 1 && (s->tuning.single_kernel == 2 || (size_t)d->width * d->height * (size_t)lf > kTwoKernelMinSamples);
    const size_t n_tiles = (size_t)((d->width + 7) / 8) * (size_t)((d->height + 7) / 8);
    const size_t state_per_frame = want_seg ? n_tiles * kWave * 4 * sizeof(float4) : 0;
    while (lf > 1 && (per_frame + state_per_frame) * lf > kMaxSampleBytes) lf >>= 1;  // (as scene_workspace does)
    out->launch_frames = lf;
    out->unit_frames = effective_unit_frames(s, d);
    out->two_kernels = want_seg ? 1 : 0;
    out->overlapped = launch_overlaps(s, d, lf) ? 1 : 0;
    out->screen_coverage = screen_coverage(s, d);
    return RT_OK;
}

int rt_render_device(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, float* accum_dev, void* stream) {
    int rc = check_render_args(s, d, frame_begin, frame_count, accum_dev);
    if (rc != RT_OK) return rc;
    if (frame_count == 0) return RT_OK;
    GUARD_DEVICE(s);
    // Overlapped launches put the path-loop kernel on one of the two internal (priority) streams and the fold on the caller's,
    // each waiting for the other's event.  Measured on MI355X / ROCm 7.2 (tools/stream_probe*.py): driven from ONE caller
    // stream that chain runs (8 live scenes, thousands of launches); as soon as overlapped launches come from a second and third
    // caller stream — even one after the other, each finished before the next starts — an internal stream stops making
    // progress, and a caller that brings priority streams competes for the same small per-priority hardware-queue pools.  So:
    //   * the first stream that renders on a device owns overlapping there; once any other stream renders on that device,
    //     overlapping is off there for the rest of the process (for everybody: mixing overlapped and plain launches across
    //     streams stalls too, and so does handing the internal streams to a second caller stream even on an idle device);
    //   * a priority stream gets its short launches on that stream itself, one after the other.
    // Without overlap (the behaviour of overlap_frames = 1) the library adds no queue of its own to the caller's.
    const int lf = effective_launch_frames(s, d) < frame_count ? effective_launch_frames(s, d) : frame_count;
    bool overlap = false;
    if (s->device >= 0 && s->device < kMaxDevices) {
        std::lock_guard<std::mutex> lock(g_overlap_mutex);
        OverlapOwner& g = g_overlap[s->device];
        if (!g.has_owner) g.owner = (hipStream_t)stream, g.has_owner = true;
        if (g.owner != (hipStream_t)stream) g.disabled = true;
        overlap = !g.disabled && launch_overlaps(s, d, lf);
    }
    if (overlap && stream) {
        int prio = 0;
        if (hipStreamGetPriority((hipStream_t)stream, &prio) != hipSuccess || prio != 0) overlap = false;
        (void)hipGetLastError();
    }
    if (overlap) {
        // A chain of overlapped launches settles into one of two steady patterns, depending on which of the two internal streams
        // (slot 0: low priority, slot 1: high) takes the first launch after the device was idle — and they differ: the 10 M-triangle
        // scene at 4K, 8 frames per call, 64 calls: 1170 Mpaths/s when the chain starts on slot 0, 1115 on slot 1; config 5, 8 frames
        // per call: 2.98 / 2.93 ms per call, 32: 11.26 / 11.03 (measured, both directions reproducible; the mechanism is the
        // dispatcher's, not ours).  So a chain that starts on an idle pipeline starts on the slot that suits the scene.
        bool idle = true;
        for (int k = 0; k < 2; ++k)
            if (s->slot[k].fold_recorded && hipEventQuery(s->slot[k].fold_done) != hipSuccess) idle = false;
        (void)hipGetLastError();  // (hipErrorNotReady is an answer, not an error)
        if (idle) s->next_slot = s->has_entry ? 0u : 1u;
    }
    Workspace ws;
    rc = scene_workspace(s, d, frame_count, &ws, overlap);
    if (rc != RT_OK) return rc;
    return launch_render<false>(s, d, frame_begin, frame_count, accum_dev, nullptr, (hipStream_t)stream, ws);
}

int rt_render_device_noalloc(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, float* accum_dev, void* stream) {
    int rc = check_render_args(s, d, frame_begin, frame_count, accum_dev);
    if (rc != RT_OK) return rc;
    if (frame_count == 0) return RT_OK;
    GUARD_DEVICE(s);
    unsigned char* base = nullptr;
    rc = scratch_address(s->device, &base);  // looked up at rt_init / rt_scene_create; here only if neither ran on this device
    if (rc != RT_OK) return rc;
    Workspace ws;  // one slot, everything on the caller's stream
    ws.slot[0].queue = (unsigned*)(base + kScratchQueueOff);
    ws.clk = (unsigned long long*)(base + kScratchClkOff);
    ws.slot[0].sobol = (float2*)(base + kScratchSobolOff);
    size_t off = kScratchBodyOff;
    const size_t threads = (size_t)render_blocks(s) * kWave;
    const size_t spill = spill_bytes_needed(s, threads);
    if (spill > kScratchBytes - off)
        return set_error(RT_ERR_INVALID_ARG, "rt_render_device_noalloc: a BVH of depth %d needs %zu bytes of stack spill; the static scratch holds %zu",
                         s->max_stack - 1, spill, kScratchBytes - off);
    if (spill) {
        ws.slot[0].spill = (int*)(base + off), ws.spill_stride = (long long)threads;
        off += (spill + 255) & ~(size_t)255;
    }
    const size_t per_frame = (size_t)d->width * d->height * 3 * sizeof(float);
    const size_t room = kScratchBytes > off ? kScratchBytes - off : 0;
    if (per_frame > room)
        return set_error(RT_ERR_INVALID_ARG, "rt_render_device_noalloc: one %dx%d frame of samples is %zu bytes; the static scratch has room for %zu (limit: width*height <= %zu pixels on this scene)",
                         d->width, d->height, per_frame, room, room / (3 * sizeof(float)));
    size_t lf = room / per_frame;
    if (lf > (size_t)effective_launch_frames(s, d)) lf = (size_t)effective_launch_frames(s, d);
    if (lf > (size_t)frame_count) lf = (size_t)frame_count;
    ws.slot[0].samples = (float*)(base + off);
    ws.launch_frames = (int)lf;
    ws.timed = false;  // hipEventCreate allocates
    // one call at a time per GPU: the whole enqueue under the lock, the device side through the guard event
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    const int dv = s->device;
    if (g_scratch_used[dv] && g_scratch_stream[dv] != (hipStream_t)stream) {
        const hipError_t q = hipEventQuery(g_scratch_done[dv]);
        if (q == hipErrorNotReady)
            return set_error(RT_ERR_INVALID_ARG, "rt_render_device_noalloc: the static scratch of device %d is still in use by a call on another stream (calls on one GPU share it: serialise them)", dv);
        if (q != hipSuccess) return set_error(RT_ERR_HIP, "rt_render_device_noalloc: hipEventQuery: %s", hipGetErrorString(q));
    }
    rc = launch_render<false>(s, d, frame_begin, frame_count, accum_dev, nullptr, (hipStream_t)stream, ws);
    if (rc != RT_OK) return rc;
    HIPCHK(hipEventRecord(g_scratch_done[dv], (hipStream_t)stream));
    g_scratch_used[dv] = true, g_scratch_stream[dv] = (hipStream_t)stream;
    return RT_OK;
}

int rt_render(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, float* accum_host) {
    int rc = check_desc(s, d, "rt_render");
    if (rc != RT_OK) return rc;
    if (!accum_host) return set_error(RT_ERR_INVALID_ARG, "rt_render: accum is NULL");
    GUARD_DEVICE(s);
    const size_t bytes = (size_t)d->width * d->height * 3 * sizeof(float);
    if (bytes > s->frame_bytes) {  // the device copy of lastFrame is kept with the scene: later calls allocate nothing
        (void)hipFree(s->d_frame);
        s->d_frame = nullptr, s->frame_bytes = 0;
        HIPCHK(hipMalloc((void**)&s->d_frame, bytes));
        s->frame_bytes = bytes;
    }
    HIPCHK(hipMemcpyAsync(s->d_frame, accum_host, bytes, hipMemcpyHostToDevice, nullptr));
    rc = rt_render_device(s, d, frame_begin, frame_count, s->d_frame, nullptr);
    if (rc != RT_OK) return rc;
    HIPCHK(hipMemcpyAsync(accum_host, s->d_frame, bytes, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    return RT_OK;
}

static int render_stats_impl(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, rt_trace_stats* out, int executed, bool camera_only = false) {
    if (!out) return set_error(RT_ERR_INVALID_ARG, "rt_render_stats: out is NULL");
    const size_t bytes = (s && d && d->width > 0 && d->height > 0) ? (size_t)d->width * d->height * 3 * sizeof(float) : 0;
    float dummy = 0.0f;
    int rc = check_render_args(s, d, frame_begin, frame_count, &dummy);
    if (rc != RT_OK) return rc;
    GUARD_DEVICE(s);
    float* dev = nullptr;
    unsigned long long* st = nullptr;
    HIPCHK(hipMalloc((void**)&dev, bytes));
    hipError_t e = hipMalloc((void**)&st, 26 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(dev, 0, bytes);
    if (e == hipSuccess) e = hipMemset(st, 0, 26 * sizeof(unsigned long long));
    unsigned long long h[26] = {0};
    if (e == hipSuccess && frame_count > 0) {
        Workspace ws;
        rc = scene_workspace(s, d, frame_count, &ws, false, camera_only);
        ws.camera_only = camera_only;
        if (rc == RT_OK) rc = launch_render<true>(s, d, frame_begin, frame_count, dev, st, nullptr, ws, executed);
        if (rc == RT_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(dev), (void)hipFree(st);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return set_error(RT_ERR_HIP, "rt_render_stats: %s", hipGetErrorString(e));
    if (getenv("RT_AMD_DIAG")) {  // diagnostics print only; no effect on what is computed
        static const char* sn[7] = {"BRDF eval", "env lookup", "finish hit", "bounce start", "BRDF sampling", "end of sample", "ray prologue"};
        for (int k = 0; k < 7; ++k)
            fprintf(stderr, "[rt diag] shade step %-14s mean lanes %.1f\n", sn[k], h[10] ? (double)h[18 + k] / (double)h[10] : 0.0);
        static const char* nm[6] = {"OP_INT", "OP_LEAF (lanes)", "OP_H(shade)", "all passes: DONE", "-", "leaf items"};
        for (int k = 0; k < 6; ++k)
            fprintf(stderr, "[rt diag] %-18s wave passes %llu, mean active lanes %.1f\n", nm[k], h[6 + 2 * k], h[6 + 2 * k] ? (double)h[7 + 2 * k] / (double)h[6 + 2 * k] : 0.0);
    }
    out->paths = h[0], out->rays = h[1], out->internal_nodes = h[2], out->leaf_nodes = h[3];
    out->triangles_tested = h[4], out->hit_updates = h[5];
    return RT_OK;
}

int rt_render_multi(rt_scene* const* scenes, int n_scenes, const rt_render_desc* d, int frame_begin, int frames_each, float* sum_host) {
    // One process, several GPUs (SURVEY 8e with the ranks folded into one caller): scene k — a replica on its own device —
    // renders frames [frame_begin + k * frames_each, + frames_each) into its own SUM buffer; all devices run at once (the
    // launches are asynchronous), then the buffers are added into sum_host in scene order (deterministic fp32 order).
    if (!scenes || n_scenes <= 0 || !d || !sum_host) return set_error(RT_ERR_INVALID_ARG, "rt_render_multi: bad arguments");
    if (frames_each < 0 || frame_begin < 0) return set_error(RT_ERR_INVALID_ARG, "rt_render_multi: negative frame range");
    if ((long long)frames_each * n_scenes > 0x7fffffffll - frame_begin) return set_error(RT_ERR_INVALID_ARG, "rt_render_multi: the frame range runs past 2^31 - 1");
    rt_render_desc ds = *d;
    ds.accumulate = RT_ACCUM_SUM;
    const size_t n = (size_t)(d->width > 0 ? d->width : 0) * (size_t)(d->height > 0 ? d->height : 0) * 3;
    const size_t bytes = n * sizeof(float);
    for (int k = 0; k < n_scenes; ++k) {
        rt_scene* s = scenes[k];
        int rc = check_render_args(s, &ds, frame_begin, frames_each, sum_host);
        if (rc != RT_OK) return rc;
        for (int j = 0; j < k; ++j)
            if (scenes[j] == s) return set_error(RT_ERR_INVALID_ARG, "rt_render_multi: scene %d appears twice (each replica owns one workspace)", k);
    }
    if (frames_each == 0) return RT_OK;
    for (int k = 0; k < n_scenes; ++k) {  // enqueue everywhere first: the devices work concurrently
        rt_scene* s = scenes[k];
        GUARD_DEVICE(s);
        if (bytes > s->frame_bytes) {
            (void)hipFree(s->d_frame);
            s->d_frame = nullptr, s->frame_bytes = 0;
            HIPCHK(hipMalloc((void**)&s->d_frame, bytes));
            s->frame_bytes = bytes;
        }
        if (bytes > s->stage_bytes) {  // pinned staging kept with the replica: the download is asynchronous and at link rate
            if (s->h_stage) (void)hipHostFree(s->h_stage);
            s->h_stage = nullptr, s->stage_bytes = 0;
            HIPCHK(hipHostMalloc((void**)&s->h_stage, bytes, hipHostMallocDefault));
            s->stage_bytes = bytes;
        }
        HIPCHK(hipMemsetAsync(s->d_frame, 0, bytes, nullptr));
        const int rc = rt_render_device(s, &ds, frame_begin + k * frames_each, frames_each, s->d_frame, nullptr);
        if (rc != RT_OK) return rc;
        HIPCHK(hipMemcpyAsync(s->h_stage, s->d_frame, bytes, hipMemcpyDeviceToHost, nullptr));  // behind the fold, on the same stream
    }
    for (int k = 0; k < n_scenes; ++k) {  // the "reduce": in scene order (per pixel the same fp32 additions in the same order as a scalar loop)
        rt_scene* s = scenes[k];
        GUARD_DEVICE(s);
        HIPCHK(hipStreamSynchronize(nullptr));
        add_into(sum_host, s->h_stage, n);  // all host cores (host_scene.cpp)
    }
    return RT_OK;
}

int rt_render_stats(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, rt_trace_stats* out) {
    return render_stats_impl(s, d, frame_begin, frame_count, out, 0);
}
int rt_render_stats_executed(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, rt_trace_stats* out) {
    return render_stats_impl(s, d, frame_begin, frame_count, out, 1);
}
int rt_render_stats_camera(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, int executed, rt_trace_stats* out) {
    return render_stats_impl(s, d, frame_begin, frame_count, out, executed ? 1 : 0, true);
}

int rt_primary_hits_device(rt_scene* s, const rt_render_desc* d, int32_t* tri_dev, float* dist_dev, void* stream) {
    int rc = check_desc(s, d, "rt_primary_hits");
    if (rc != RT_OK) return rc;
    if (!tri_dev || !dist_dev) return set_error(RT_ERR_INVALID_ARG, "rt_primary_hits: NULL output");
    GUARD_DEVICE(s);
    RenderParams P;
    fill_params(s, d, &P);
    const int blocks = ((d->width + 15) / 16) * ((d->height + 15) / 16);  // 256 threads per 16x16 tile
    rc = grow_scene_spill(s, (size_t)blocks * kBlock, &P.spill, &P.spill_stride);
    if (rc != RT_OK) return rc;
    P.hit_tri = tri_dev, P.hit_dist = dist_dev;
    TimedLaunch tl;
    HIPCHK(hipEventCreate(&tl.a));
    HIPCHK(hipEventCreate(&tl.b));
    HIPCHK(hipEventRecord(tl.a, (hipStream_t)stream));
    if (P.spill)
        hipLaunchKernelGGL((primary_kernel<false, true>), dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, P);
    else
        hipLaunchKernelGGL((primary_kernel<false, false>), dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, P);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(tl.b, (hipStream_t)stream));
    remember_launch(tl);
    return RT_OK;
}

int rt_primary_hits(rt_scene* s, const rt_render_desc* d, int32_t* tri_host, float* dist_host) {
    int rc = check_desc(s, d, "rt_primary_hits");
    if (rc != RT_OK) return rc;
    if (!tri_host || !dist_host) return set_error(RT_ERR_INVALID_ARG, "rt_primary_hits: NULL output");
    GUARD_DEVICE(s);
    const size_t n = (size_t)d->width * d->height;
    int32_t* dt = nullptr;
    float* dd = nullptr;
    HIPCHK(hipMalloc((void**)&dt, n * 4));
    hipError_t e = hipMalloc((void**)&dd, n * 4);
    if (e == hipSuccess) {
        rc = rt_primary_hits_device(s, d, dt, dd, nullptr);
        if (rc == RT_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(tri_host, dt, n * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(dist_host, dd, n * 4, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(dt), (void)hipFree(dd);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return set_error(RT_ERR_HIP, "rt_primary_hits: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_kernel_timing(rt_scene* s, rt_kernel_timing_info* out, int reset) {
    if (!s || !out) return set_error(RT_ERR_INVALID_ARG, "rt_kernel_timing: NULL argument");
    GUARD_DEVICE(s);
    double total = 0, mx = 0, first = 0;
    int n = 0, nfirst = 0;
    for (auto& tl : g_launches) {
        float ms = 0;
        if (hipEventSynchronize(tl.b) == hipSuccess && hipEventElapsedTime(&ms, tl.a, tl.b) == hipSuccess) {
            total += ms;
            if (ms > mx) mx = ms;
            ++n;
            if (tl.a0 && hipEventElapsedTime(&ms, tl.a0, tl.a) == hipSuccess) first += ms, ++nfirst;
        }
    }
    unsigned long long ticks = 0;
    unsigned long long* cell = s->d_clk;
    HIPCHK(hipMemcpy(&ticks, cell, sizeof(ticks), hipMemcpyDeviceToHost));
    out->avg_ms = n ? (float)(total / n) : 0.0f;
    out->first_avg_ms = nfirst ? (float)(first / nfirst) : 0.0f;
    out->max_ms = (float)mx;
    out->launches = n;
    // the longest-lived wave of the longest launch spans that launch: ticks / duration = the shader clock it ran at
    out->shader_clock_mhz = (mx > 0.0 && ticks) ? (float)((double)ticks / (mx * 1e3)) : 0.0f;
    if (reset) {
        HIPCHK(hipMemset(cell, 0, sizeof(ticks)));
        for (auto& tl : g_launches) destroy_timed(tl);
        g_launches.clear();
    }
    return RT_OK;
}

int rt_kernel_time_ms(float* avg_ms, int* launches, int reset) {
    double total = 0;
    int n = 0;
    for (auto& tl : g_launches) {
        float ms = 0;
        if (hipEventSynchronize(tl.b) == hipSuccess && hipEventElapsedTime(&ms, tl.a, tl.b) == hipSuccess) {
            total += ms;
            ++n;
        }
    }
    if (avg_ms) *avg_ms = n ? (float)(total / n) : 0.0f;
    if (launches) *launches = n;
#ifdef RT_SHADE_CYCLES
    {
        unsigned long long h[16] = {0}, z[16] = {0};
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_shadecycles), sizeof(h)) == hipSuccess && h[8]) {
            static const char* nm[8] = {"BRDF eval", "env lookup", "contributions", "finish hit", "bounce start", "BRDF sampling", "end + next sample", "ray prologue"};
            double tot = 0;
            for (int k = 0; k < 8; ++k) tot += (double)h[k];
            for (int k = 0; k < 8; ++k)
                fprintf(stderr, "[rt shade] %-18s %6.1f ticks per pass  %5.1f%%\n", nm[k], (double)h[k] / (double)h[8], 100.0 * h[k] / tot);
            fprintf(stderr, "[rt shade] %llu passes, %.0f ticks per pass\n", h[8], tot / (double)h[8]);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_shadecycles), z, sizeof(z));
        }
    }
#endif
#ifdef RT_OP_CYCLES
    {
        unsigned long long h[8] = {0}, z[8] = {0};
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_opcycles), sizeof(h)) == hipSuccess && h[4]) {
            const double tot = (double)h[3];
            fprintf(stderr, "[rt cycles] waves %llu, ticks/wave %.0f: INT %.1f%%  LEAF %.1f%%  shade %.1f%%  scheduler+prologue %.1f%%\n", h[4],
                    tot / (double)h[4], 100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot,
                    100.0 * (tot - h[0] - h[1] - h[2]) / tot);
            fprintf(stderr, "[rt cycles] unit acquisitions per wave %.1f, ticks each %.0f = %.2f%% of the wave's ticks\n", (double)h[6] / (double)h[4],
                    h[6] ? (double)h[5] / (double)h[6] : 0.0, 100.0 * h[5] / tot);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_opcycles), z, sizeof(z));
        }
    }
#endif
    if (reset) {
        for (auto& tl : g_launches) destroy_timed(tl);
        g_launches.clear();
    }
    return RT_OK;
}

}  // extern "C"
