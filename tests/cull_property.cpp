// Property test of the culling lemma ("Culling"), on the product's own traversal arithmetic compiled
// for the host (csrc/rt_trav.h, -ffp-contract=off):
//   for a cull-eligible ray and ANY box that contains a triangle's vertices, if hitTriangle accepts the triangle with
//   distance t then hitAABB_cull's conservative entry distance t0c of that box satisfies t0c <= t.
// Rays are aimed at the places where rounding decides: edges, vertices, just beyond the tips of needles and ulp-thin
// slivers, at grazing angles to the triangle's plane and nearly parallel to the box faces, from near and far, at tiny, unit
// and huge scales and far from the origin.  Also reports the largest fraction of the margin that any accepted hit needed
// (1 / safety factor).  Usage: cull_property [samples] [seed]; exit status 1 on any violation.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "../ray_tracing_amd/csrc/rt_trav.h"

using namespace rta;

int main(int argc, char** argv) {
    const long long samples = argc > 1 ? atoll(argv[1]) : 4000000;
    std::mt19937_64 gen(argc > 2 ? atoll(argv[2]) : 12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> G(0.0, 1.0);
    auto rnd_dir = [&](double* v) {
        double n;
        do {
            v[0] = G(gen), v[1] = G(gen), v[2] = G(gen);
            n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        } while (n < 1e-6);
        v[0] /= n, v[1] /= n, v[2] /= n;
    };
    long long accepted = 0, checked = 0, violations = 0, ineligible = 0, interior_checked = 0;
    double worst = 0.0;
    long long by_kind[4] = {0, 0, 0, 0};
    for (long long it = 0; it < samples; ++it) {
        const double scale = pow(10.0, -3.0 + 6.0 * U(gen));
        double off[3] = {0, 0, 0};
        if (U(gen) < 0.3)
            for (int c = 0; c < 3; ++c) off[c] = (U(gen) - 0.5) * 2000.0 * scale;
        // ---- a triangle ----
        const int kind = (int)(U(gen) * 4.0) & 3;  // 0 ordinary, 1 needle, 2 ulp sliver, 3 axis-aligned strip
        double P[3][3];
        for (int v = 0; v < 3; ++v)
            for (int c = 0; c < 3; ++c) P[v][c] = G(gen) * scale;
        if (kind == 1) {
            const double s = 0.2 + 2.8 * U(gen), w = pow(10.0, -7.0 + 5.0 * U(gen)) * scale;
            for (int c = 0; c < 3; ++c) P[2][c] = P[0][c] + (P[1][c] - P[0][c]) * s + G(gen) * w;
        } else if (kind == 3) {
            const int ax = (int)(U(gen) * 3.0) % 3, a2 = (ax + 1) % 3, a3 = (ax + 2) % 3;
            const double x0 = G(gen) * scale, y0 = G(gen) * scale, w = pow(10.0, -5.0 + 4.0 * U(gen)) * scale;
            P[0][ax] = x0, P[0][a2] = y0, P[0][a3] = -1.5 * scale;
            P[1][ax] = x0, P[1][a2] = y0 + w, P[1][a3] = -1.5 * scale;
            P[2][ax] = x0, P[2][a2] = y0, P[2][a3] = 1.5 * scale;
        }
        float q[12];
        for (int v = 0; v < 3; ++v)
            for (int c = 0; c < 3; ++c) q[v * 3 + c] = (float)(P[v][c] + off[c]);
        if (kind == 2) {
            const int k = 1 << (int)(U(gen) * 7.0);
            for (int c = 0; c < 3; ++c) q[6 + c] = q[3 + c] * (1.0f + (float)k * 1.1920929e-7f);
        }
        const vec3 p1 = v3(q[0], q[1], q[2]), p2 = v3(q[3], q[4], q[5]), p3 = v3(q[6], q[7], q[8]);
        const vec3 N0 = normalize(cross(p2 - p1, p3 - p1));  // f5:180, as rt_scene_create stores it
        if (!finite3(N0) || (N0.x == 0.0f && N0.y == 0.0f && N0.z == 0.0f)) continue;  // never accepted
        const float eps_box = tri_cull_margin(q);
        float ext = 0.0f;
        for (int k = 0; k < 9; ++k) ext = fmaxf(ext, fabsf(q[k]));
        const float scene_eps = nextafterf(ext / 8192.0f, INFINITY);
        // ---- a target point on / near the triangle ----
        double b0 = U(gen), b1 = U(gen);
        if (b0 + b1 > 1.0) b0 = 1.0 - b0, b1 = 1.0 - b1;
        const double mode = U(gen);
        if (mode < 0.25) b1 = pow(10.0, -9.0 * U(gen)) * (U(gen) < 0.5 ? 1.0 : -1.0);       // on / across an edge
        else if (mode < 0.5) b0 = 1.0 + pow(10.0, -8.0 * U(gen)) * (U(gen) < 0.7 ? 1.0 : -1.0), b1 = pow(10.0, -9.0 * U(gen)) * (U(gen) - 0.5);  // at / beyond a tip
        const int r0 = (int)(U(gen) * 3.0) % 3, r1 = (r0 + 1) % 3, r2 = (r0 + 2) % 3;
        double T[3];
        for (int c = 0; c < 3; ++c)
            T[c] = (double)q[r0 * 3 + c] + b0 * ((double)q[r1 * 3 + c] - (double)q[r0 * 3 + c]) + b1 * ((double)q[r2 * 3 + c] - (double)q[r0 * 3 + c]);
        // ---- a ray towards it ----
        double dir[3];
        rnd_dir(dir);
        const double rm = U(gen);
        if (rm < 0.35) {  // grazing: almost inside the triangle's plane
            const double g = pow(10.0, -6.0 + 5.0 * U(gen));
            double e[3] = {(double)q[3] - q[0], (double)q[4] - q[1], (double)q[5] - q[2]};
            const double n = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + 1e-300;
            double f[3] = {(double)q[6] - q[0], (double)q[7] - q[1], (double)q[8] - q[2]};
            const double m = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) + 1e-300;
            const double ca = G(gen), cb = G(gen);
            for (int c = 0; c < 3; ++c) dir[c] = ca * e[c] / n + cb * f[c] / m + g * (double)(c == 0 ? N0.x : (c == 1 ? N0.y : N0.z));
        } else if (rm < 0.6) {  // nearly parallel to a box face
            const int ax = (int)(U(gen) * 3.0) % 3;
            dir[ax] *= pow(10.0, -9.0 * U(gen));
        }
        const double dn = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        if (!(dn > 0.0)) continue;
        const double dist = scale * pow(10.0, -2.0 + 5.0 * U(gen));
        const vec3 S = v3((float)(T[0] - dir[0] / dn * dist), (float)(T[1] - dir[1] / dn * dist), (float)(T[2] - dir[2] / dn * dist));
        const vec3 d = normalize(v3((float)(T[0] - (double)S.x), (float)(T[1] - (double)S.y), (float)(T[2] - (double)S.z)));
        const vec3 invdir = v3(1.0f) / d;
        const bool slow = !(finite3(invdir) && finite3(S));
        if (slow) continue;
        float t;
        if (!hitTriangleT(p1, p2, p3, N0, S, d, &t)) continue;
        ++accepted, ++by_kind[kind];
        const float eps_ray = cull_margin(scene_eps, eps_box, S, invdir, false);
        if (!(eps_ray < INFINITY) || !(eps_box < INFINITY)) {
            ++ineligible;  // never culled
            continue;
        }
        // ---- short climb: a box that contains the triangle with a margin of mu = scene margin / 2 + shape
        // margin on all six sides passes the REFERENCE's own box test for this ray, for certain (rays with eps_ray <= 4 * scene) ----
        if (eps_ray <= 4.0f * scene_eps) {
            const float mu = 0.5f * scene_eps + eps_box;  // = climb_eps / 8 + shape margin (host_prep.cpp)
            vec3 A = min3(p1, min3(p2, p3)), B = max3(p1, max3(p2, p3));
            for (int rep = 0; rep < 3; ++rep) {
                const float g = mu * (1.0f + (float)(rep * rep) * (float)U(gen) * 50.0f);
                // grow by at least mu: round the grown faces outwards
                const vec3 A2 = v3(nextafterf(A.x - g, -INFINITY), nextafterf(A.y - g, -INFINITY), nextafterf(A.z - g, -INFINITY));
                const vec3 B2 = v3(nextafterf(B.x + g, INFINITY), nextafterf(B.y + g, INFINITY), nextafterf(B.z + g, INFINITY));
                ++interior_checked;
                if (!(hitAABB(S, invdir, A2, B2) > 0.0f)) {
                    if (++violations <= 10) fprintf(stderr, "VIOLATION kind %d: interior box refused by the reference's hitAABB, t %.9g mu %g\n", kind, t, mu);
                }
            }
        }
        // ---- boxes that contain the triangle: the tight one and randomly grown ones (ancestors) ----
        vec3 AA = min3(p1, min3(p2, p3)), BB = max3(p1, max3(p2, p3));
        for (int rep = 0; rep < 4; ++rep) {
            float t0c, t1c;
            (void)hitAABB_cull(S, invdir, AA, BB, eps_ray, eps_box, &t0c, &t1c);
            ++checked;
            // the entry BVH's conservative test must let this box through for any best >= t
            if (!(t0c <= t) || !(t <= t1c) || !(hitAABB_entry(S, invdir, AA, BB, eps_ray, eps_box, t) > 0.0f)) {
                if (++violations <= 10)
                    fprintf(stderr, "VIOLATION kind %d: t0c %.9g <= t %.9g <= t1c fails  S (%g %g %g) d (%g %g %g) eps %g + %g\n", kind, t0c, t, S.x, S.y, S.z, d.x, d.y, d.z, eps_ray, eps_box);
            }
            // the margin this hit needed on the binding axis, as a fraction of the margin provided
            const float inv[3] = {invdir.x, invdir.y, invdir.z}, a[3] = {AA.x, AA.y, AA.z}, b[3] = {BB.x, BB.y, BB.z}, s[3] = {S.x, S.y, S.z};
            for (int c = 0; c < 3; ++c) {
                const float f = (b[c] - s[c]) * inv[c], n = (a[c] - s[c]) * inv[c];
                const double need = ((double)fminf(f, n) - (double)t) / fabs((double)inv[c]);
                const double frac = need / ((double)eps_ray + (double)eps_box);
                if (frac > worst) worst = frac;
            }
            const float grow = (float)(scale * pow(10.0, -4.0 + 5.0 * U(gen)));
            AA = AA - v3((float)U(gen) * grow, (float)U(gen) * grow, (float)U(gen) * grow);
            BB = BB + v3((float)U(gen) * grow, (float)U(gen) * grow, (float)U(gen) * grow);
        }
    }
    printf("cull_property: %lld samples, %lld accepted hits (ordinary %lld, needle %lld, sliver %lld, strip %lld), %lld never-cull, %lld box checks, %lld interior-box checks, "
           "%lld violations, largest margin fraction needed %.4f (safety factor %.1f)\n",
           samples, accepted, by_kind[0], by_kind[1], by_kind[2], by_kind[3], ineligible, checked, interior_checked, violations, worst, worst > 0 ? 1.0 / worst : 0.0);
    return violations ? 1 : 0;
}
