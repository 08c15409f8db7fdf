// Traversal arithmetic of the path loop: hitAABB (T2), hitTriangle's distance part (T1) and the culling test, as
// host+device inline code (the HIP kernels in rt_device.hip are the only product callers; tests/cull_property.cpp
// compiles the same functions for the host to check the culling lemma on adversarial rays).  -ffp-contract=off.
#pragma once
#include "rt_shade.h"

namespace rta {

// ---- T2: hitAABB, f5:225-238 ------------------------------------------------------------------------------------
RT_HD float hitAABB(vec3 S, vec3 invdir, vec3 AA, vec3 BB) {
    const vec3 f = (BB - S) * invdir;
    const vec3 n = (AA - S) * invdir;
    const vec3 tmax = max3(f, n);
    const vec3 tmin = min3(f, n);
    const float t1 = fmin_(tmax.x, fmin_(tmax.y, tmax.z));
    const float t0 = fmax_(tmin.x, fmax_(tmin.y, tmin.z));
    return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}

// ---- culling against the closest hit so far -------------------------------------------------------------------------
// The reference never culls (f5:295-307): it walks every node whose box the ray hits.  A subtree can be skipped without
// changing any result when none of its triangles could still be accepted, i.e. when every accepted hit distance t in it
// satisfies t > best (acceptance needs t < best, f5:278; chapter 2: t <= best).  hitAABB_cull returns, beside the
// reference's d, a conservative entry distance t0c of the box INFLATED by `eps` in space, evaluated per axis:
//     t0c = max_i ( min(n_i, f_i) - eps * |invdir_i| )
// DESIGN.md section 5 ("Culling") proves: for a cull-eligible ray (finite S and invdir, eps * max|invdir| < 1e30) and a
// validated BVH (every box contains its children's boxes and its triangles' vertices), any triangle below the box that
// hitTriangle accepts with distance t has t0c <= t.  So `t0c > best` skips only subtrees that cannot change the result.
// eps = eps_ray + eps_box: eps_ray = 2^-13 * (max|S_i| + scene extent); eps_box (kept per child in the node record) =
// 2^-14 * max over the triangles below the box of longest edge / sin(smallest angle / 2).  Together >= 20x the distance
// by which hitTriangle's point S + d*t can miss the triangle through rounding; applied per axis, so rays nearly parallel
// to a box face are covered (the margin is eps / |d_i| on axis i, not a scalar).
// d itself is hitAABB (f5:225-238) with hardware min/max: equal to the ternary form whenever no operand is NaN (the
// results are only compared, so the sign of a zero does not matter); rays that can produce NaN (a zero/NaN direction
// component or a non-finite origin) take the exact form instead (`slow`) and are never culled.
constexpr float kCullRel = 1.0f / 8192.0f;  // 2^-13
// t1c (optional) is the matching conservative EXIT distance, min_i ( max(n_i, f_i) + eps * |invdir_i| ): by the same
// argument t <= t1c for every accepted hit below the box.  The entry BVH (a device-private tree that only supplies a
// distance bound, see rt_device.hip) is walked with the conservative test alone: a box may hold an acceptable hit closer
// than `best` only if t0c <= t1c, t1c >= 0 and t0c <= best.
RT_HD float hitAABB_cull(vec3 S, vec3 invdir, vec3 AA, vec3 BB, float eps_ray, float eps_box, float* t0c, float* t1c = nullptr) {
    const float eps = eps_ray + eps_box;
    const vec3 f = (BB - S) * invdir;
    const vec3 n = (AA - S) * invdir;
    const float mx = __builtin_fminf(f.x, n.x), my = __builtin_fminf(f.y, n.y), mz = __builtin_fminf(f.z, n.z);
    const float Mx = __builtin_fmaxf(f.x, n.x), My = __builtin_fmaxf(f.y, n.y), Mz = __builtin_fmaxf(f.z, n.z);
    const float t1 = __builtin_fminf(__builtin_fminf(Mx, My), Mz);
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(mx, my), mz);
    const float ex = eps * __builtin_fabsf(invdir.x), ey = eps * __builtin_fabsf(invdir.y), ez = eps * __builtin_fabsf(invdir.z);
    *t0c = __builtin_fmaxf(__builtin_fmaxf(mx - ex, my - ey), mz - ez);
    if (t1c) *t1c = __builtin_fminf(__builtin_fminf(Mx + ex, My + ey), Mz + ez);
    return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}
// The entry BVH's box test: > 0 (a key to order the two children, near first) when the box may hold an acceptable hit
// closer than `best`, else -1.
RT_HD float hitAABB_entry(vec3 S, vec3 invdir, vec3 AA, vec3 BB, float eps_ray, float eps_box, float best) {
    float t0c, t1c;
    (void)hitAABB_cull(S, invdir, AA, BB, eps_ray, eps_box, &t0c, &t1c);
    const bool may = (t0c <= t1c) && (t1c >= 0.0f) && (t0c <= best);
    return may ? __builtin_fmaxf(t0c, 1e-30f) : -1.0f;
}
// the culling margin of a ray (+inf: never cull): `slow` rays (NaN-exact box test) and rays whose margin could overflow
RT_HD float cull_margin(float scene_eps, float geo_max, vec3 S, vec3 invdir, bool slow) {
    const float inf = __builtin_inff();
    const float m = __builtin_fmaxf(__builtin_fmaxf(fabs_(S.x), fabs_(S.y)), fabs_(S.z));
    const float im = __builtin_fmaxf(__builtin_fmaxf(fabs_(invdir.x), fabs_(invdir.y)), fabs_(invdir.z));
    const float e = m * kCullRel + scene_eps;
    return (!slow && (e + geo_max) * im < 1e30f) ? e : inf;
}

RT_HD bool finite3(vec3 v) {
    const float inf = __builtin_inff();
    return (fabs_(v.x) < inf) && (fabs_(v.y) < inf) && (fabs_(v.z) < inf);
}

// ---- T1: hitTriangle (distance part), f5:168-209.  N0 = precomputed normalize(cross(p2-p1, p3-p1)). -------------
// `t_max` (callers that keep a closest hit): a hit at t > t_max could not be accepted by the caller anyway (acceptance needs
// t < best, f5:278; chapter 2: t <= best, c2:512), so the inside test is skipped for it — same accept / reject for every t the
// caller would use; a wave whose lanes all see a triangle beyond their closest hit (coherent camera rays) skips 2/3 of the test.
RT_HD bool hitTriangleT(vec3 p1, vec3 p2, vec3 p3, vec3 N0, vec3 S, vec3 d, float* t_out, float t_max = __builtin_inff()) {
    // The reference flips N to face the ray first (f5:183-186).  Every later use is odd in N (|dot(N,d)|, the quotient
    // of two dot products with N, the signs of a,b,c tested for "all > 0 or all < 0"), and fp32 negation commutes with
    // round-to-nearest products and sums, so the unflipped normal gives bit-identical t and the same accept/reject.
    const vec3 N = N0;
    if (fabs_(dot(N, d)) < 0.00001f) return false;
    const float t = (dot(N, p1) - dot(S, N)) / dot(d, N);
    if (t < 0.0005f) return false;
    if (t > t_max) return false;
    const vec3 P = S + d * t;
    const vec3 c1 = cross(p2 - p1, P - p1);
    const vec3 c2 = cross(p3 - p2, P - p2);
    const vec3 c3 = cross(p1 - p3, P - p3);
    const float a = dot(c1, N), b = dot(c2, N), c = dot(c3, N);
    const bool r1 = (a > 0.0f && b > 0.0f && c > 0.0f);
    const bool r2 = (a < 0.0f && b < 0.0f && c < 0.0f);
    if (!(r1 || r2)) return false;
    *t_out = t;
    return true;
}


// Shape term of the culling margin of one triangle (9 floats p1 p2 p3), host side, in double precision:
// 2^-14 * longest edge / sin(smallest angle / 2), rounded up to fp32 (+inf for collinear vertices).  DESIGN.md section 5.
inline float tri_cull_margin(const float* q) {
    const float inf = __builtin_inff();
    double P[3][3];
    for (int v = 0; v < 3; ++v)
        for (int c = 0; c < 3; ++c) P[v][c] = q[v * 3 + c];
    double len[3], lmax = 0.0, smin = 1.0;
    for (int v = 0; v < 3; ++v) {  // edge v: P[v] -> P[(v+1)%3]
        const double* a = P[v];
        const double* b = P[(v + 1) % 3];
        len[v] = sqrt((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]) + (b[2] - a[2]) * (b[2] - a[2]));
        if (len[v] > lmax) lmax = len[v];
    }
    for (int v = 0; v < 3; ++v) {  // angle at P[v], between edges v and (v+2)%3
        const double* a = P[v];
        const double* b = P[(v + 1) % 3];
        const double* c = P[(v + 2) % 3];
        const double l1 = len[v], l2 = len[(v + 2) % 3];
        double sh = 0.0;  // sin(angle / 2)
        if (l1 > 0.0 && l2 > 0.0) {
            const double cosang = ((b[0] - a[0]) * (c[0] - a[0]) + (b[1] - a[1]) * (c[1] - a[1]) + (b[2] - a[2]) * (c[2] - a[2])) / (l1 * l2);
            const double h = 0.5 * (1.0 - cosang);
            sh = h > 0.0 ? sqrt(h) : 0.0;
        }
        if (sh < smin) smin = sh;
    }
    const double g = smin > 0.0 ? lmax / smin / 16384.0 : (double)inf;  // 2^-14 * geo
    return g < 1e30 ? nextafterf((float)g, inf) : inf;
}

}  // namespace rta
