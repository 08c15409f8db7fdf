// Chapter 1 (config 1, host only — "no GPU"): the CPU Cornell-box path tracer of path_tracing_1chap/path_tracing.cpp
// (c1), restated race-free.  Arithmetic types follow the reference (float vec3, double material scalars, double image);
// the scene is the literal of c1:355-400.  Differences, on purpose: the reference's OpenMP loop shares one mt19937 and
// adds into the image without atomics from 50 threads (c1:227-233, 407-475 — data races); here rows are independent
// work items with one mt19937 each, seeded from (seed, row), so the result is deterministic for any thread count.
// The reference seeds from std::random_device, so only statistical parity with its renders is possible.
#include <math.h>
#include <string.h>

#include <random>
#include <vector>

#include "rt_host.h"

namespace rta {
namespace chap1 {

struct Ray {
    vec3 startpoint = v3(0), direction = v3(0);
};
struct Material {  // c1:46-56
    bool isEmissive = false;
    vec3 normal = v3(0);
    vec3 color = v3(0);
    double specularRate = 0.0f;
    double roughness = 1.0f;
    double refractRate = 0.0f;
    double refractAngle = 1.0f;
    double refractRoughtness = 0.0f;
};
struct HitResult {  // c1:58-64
    bool isHit = false;
    double distance = 0.0f;
    vec3 HitPoint = v3(0);
    Material material;
};
struct Shape {
    int kind = 0;  // 0 triangle, 1 sphere
    vec3 P1, P2, P3;
    vec3 O;
    double R = 0;
    Material material;
};

static Shape make_triangle(vec3 p1, vec3 p2, vec3 p3, vec3 C) {  // c1:82-87
    Shape s;
    s.kind = 0, s.P1 = p1, s.P2 = p2, s.P3 = p3;
    s.material.normal = normalize(cross(p2 - p1, p3 - p1));
    s.material.color = C;
    return s;
}
static Shape make_sphere(vec3 o, double r, vec3 c) {  // c1:131
    Shape s;
    s.kind = 1, s.O = o, s.R = r, s.material.color = c;
    return s;
}

static HitResult intersect_triangle(const Shape& s, const Ray& ray) {  // c1:94-124
    HitResult res;
    const vec3 S = ray.startpoint, d = ray.direction, N = s.material.normal;
    if (fabsf(dot(N, d)) < 0.00001f) return res;
    const float t = (dot(N, s.P1) - dot(S, N)) / dot(d, N);
    if (t < 0.0005f) return res;
    const vec3 P = S + d * t;
    const vec3 c1 = cross(s.P2 - s.P1, P - s.P1), c2 = cross(s.P3 - s.P2, P - s.P2), c3 = cross(s.P1 - s.P3, P - s.P3);
    const vec3 n = s.material.normal;
    if (dot(c1, n) < 0 || dot(c2, n) < 0 || dot(c3, n) < 0) return res;
    res.isHit = true;
    res.distance = t;
    res.HitPoint = P;
    res.material = s.material;
    res.material.normal = N;
    return res;
}

static HitResult intersect_sphere(const Shape& s, const Ray& ray) {  // c1:136-173
    HitResult res;
    const vec3 S = ray.startpoint, d = ray.direction;
    const vec3 OS3 = s.O - S;
    const float OS = sqrtf(dot(OS3, OS3));
    const float SH = dot(OS3, d);
    const float OH = (float)sqrt(pow((double)OS, 2) - pow((double)SH, 2));
    if (OH > s.R) return res;
    const float PH = (float)sqrt(pow(s.R, 2) - pow((double)OH, 2));
    const float t1 = SH - PH, t2 = SH + PH;
    float t;
    if (t1 < 0 && t2 < 0)
        return res;
    else if (t1 > 0 && t2 > 0)
        t = t1 < t2 ? t1 : t2;
    else
        t = t1 > t2 ? t1 : t2;
    const vec3 P = S + t * d;
    if (fabsf(t1) < 0.0005f || fabsf(t2) < 0.0005f) return res;
    res.HitPoint = P;
    res.isHit = true;
    res.distance = t;
    res.material = s.material;
    res.material.normal = normalize(P - s.O);
    return res;
}

struct Tracer {
    std::vector<Shape> shapes;
    std::mt19937 gen;
    std::uniform_real_distribution<> dis{0.0, 1.0};
    double randf() { return dis(gen); }

    HitResult shoot(const Ray& ray) const {  // c1:214-225
        HitResult res, r;
        res.distance = 1145141919.810f;
        for (const Shape& s : shapes) {
            r = s.kind == 0 ? intersect_triangle(s, ray) : intersect_sphere(s, ray);
            if (r.isHit && r.distance < res.distance) res = r;
        }
        return res;
    }
    vec3 randomvec3() {  // c1:237-249
        vec3 d;
        do {
            const float a = (float)randf(), b = (float)randf(), c = (float)randf();
            d = 2.0f * v3(a, b, c) - v3(1, 1, 1);
        } while (dot(d, d) > 1.0);
        return normalize(d);
    }
    vec3 randomDirection(vec3 n) { return normalize(randomvec3() + n); }

    static vec3 mixd(vec3 x, vec3 y, double a) {  // glm::mix(vec3, vec3, double): evaluated in double
        return v3((float)((double)x.x * (1.0 - a) + (double)y.x * a), (float)((double)x.y * (1.0 - a) + (double)y.y * a),
                  (float)((double)x.z * (1.0 - a) + (double)y.z * a));
    }
    static vec3 refract(vec3 I, vec3 N, float eta) {  // glm::refract
        const float dotValue = dot(N, I);
        const float k = 1.0f - eta * eta * (1.0f - dotValue * dotValue);
        return (k >= 0.0f) ? (eta * I - (eta * dotValue + sqrtf(k)) * N) : v3(0);
    }

    // the three-way branch shared by main() (c1:448-468) and pathTracing (c1:301-321); `depth` < 0 = called from main
    vec3 scatter(const Ray& ray, const HitResult& res, Ray randomRay, int next_depth, float consine, bool from_main) {
        const double r = randf();
        vec3 color;
        if (r < res.material.specularRate) {
            const vec3 ref = normalize(reflect(ray.direction, res.material.normal));
            randomRay.direction = mixd(ref, randomRay.direction, res.material.roughness);
            color = pathTracing(randomRay, next_depth);
            if (!from_main) color = color * consine;
        } else if (res.material.specularRate <= r && r <= res.material.refractRate) {
            const vec3 ref = normalize(refract(ray.direction, res.material.normal, float(res.material.refractAngle)));
            randomRay.direction = mixd(ref, -randomRay.direction, res.material.refractRoughtness);
            color = pathTracing(randomRay, next_depth);
            if (!from_main) color = color * consine;
        } else {
            const vec3 srcColor = res.material.color;
            vec3 ptColor = pathTracing(randomRay, next_depth);
            if (!from_main) ptColor = ptColor * consine;
            color = from_main ? (ptColor * srcColor) : (srcColor * ptColor);
        }
        return color;
    }

    vec3 pathTracing(const Ray& ray, int depth) {  // c1:277-324
        if (depth > 8) return v3(0);
        const HitResult res = shoot(ray);
        if (!res.isHit) return v3(0);
        if (res.material.isEmissive) return res.material.color;
        const double r = randf();
        const float P = 0.8;
        if (r > P) return v3(0);
        Ray randomRay;
        randomRay.startpoint = res.HitPoint;
        randomRay.direction = randomDirection(res.material.normal);
        const float consine = fabsf(dot(-ray.direction, res.material.normal));
        const vec3 color = scatter(ray, res, randomRay, depth + 1, consine, false);
        return color / P;
    }
};

static std::vector<Shape> cornell_scene() {  // c1:355-400
    const vec3 RED = v3(1, 0.5, 0.5), GREEN = v3(0.5, 1, 0.5), BLUE = v3(0.5, 0.5, 1), YELLOW = v3(1.0, 1.0, 0.1);
    const vec3 CYAN = v3(0.1, 1.0, 1.0), WHITE = v3(1, 1, 1);
    std::vector<Shape> S;
    S.push_back(make_triangle(v3(-0.15, 0.4, -0.6), v3(-0.15, -0.95, -0.6), v3(0.15, 0.4, -0.6), YELLOW));
    S.push_back(make_triangle(v3(0.15, 0.4, -0.6), v3(-0.15, -0.95, -0.6), v3(0.15, -0.95, -0.6), YELLOW));
    Shape l1 = make_triangle(v3(0.4, 0.99, 0.4), v3(-0.4, 0.99, 0.4), v3(-0.4, 0.99, -0.4), WHITE);
    Shape l2 = make_triangle(v3(0.4, 0.99, 0.4), v3(-0.4, 0.99, -0.4), v3(0.4, 0.99, -0.4), WHITE);
    l1.material.isEmissive = true, l2.material.isEmissive = true;
    S.push_back(l1), S.push_back(l2);
    S.push_back(make_triangle(v3(1, -1, 1), v3(-1, -1, -1), v3(-1, -1, 1), WHITE));   // bottom
    S.push_back(make_triangle(v3(1, -1, 1), v3(1, -1, -1), v3(-1, -1, -1), WHITE));
    S.push_back(make_triangle(v3(1, 1, 1), v3(-1, 1, 1), v3(-1, 1, -1), WHITE));      // top
    S.push_back(make_triangle(v3(1, 1, 1), v3(-1, 1, -1), v3(1, 1, -1), WHITE));
    S.push_back(make_triangle(v3(1, -1, -1), v3(-1, 1, -1), v3(-1, -1, -1), CYAN));   // back
    S.push_back(make_triangle(v3(1, -1, -1), v3(1, 1, -1), v3(-1, 1, -1), CYAN));
    S.push_back(make_triangle(v3(-1, -1, -1), v3(-1, 1, 1), v3(-1, -1, 1), BLUE));    // left
    S.push_back(make_triangle(v3(-1, -1, -1), v3(-1, 1, -1), v3(-1, 1, 1), BLUE));
    S.push_back(make_triangle(v3(1, 1, 1), v3(1, -1, -1), v3(1, -1, 1), RED));        // right
    S.push_back(make_triangle(v3(1, -1, -1), v3(1, 1, 1), v3(1, 1, -1), RED));
    Shape sph2 = make_sphere(v3(-0.6, -0.8, -0.5), 0.2, WHITE);
    sph2.material.specularRate = 0.9, sph2.material.roughness = 0.0;
    S.push_back(sph2);
    Shape sph = make_sphere(v3(0.0, -0.3, 0.0), 0.4, WHITE);
    sph.material.specularRate = 0.1, sph.material.refractAngle = 0.1, sph.material.refractRate = 0.95;
    S.push_back(sph);
    Shape sph1 = make_sphere(v3(0.6, -0.8, -0.5), 0.2, GREEN);
    sph1.material.specularRate = 0.3, sph1.material.roughness = 0.1;
    S.push_back(sph1);
    return S;
}

}  // namespace chap1
}  // namespace rta

using namespace rta;

extern "C" int rt_chap1_render(int width, int height, int samples, unsigned seed, double* image) {
    if (width <= 0 || height <= 0 || samples <= 0 || !image) return set_error(RT_ERR_INVALID_ARG, "rt_chap1_render: bad arguments");
    const std::vector<chap1::Shape> shapes = chap1::cornell_scene();
    const double BRIGHTNESS = (2.0f * 3.1415926f) * (1.0f / double(samples));  // c1:14
    const double SCREEN_Z = 1.1;                                                 // c1:19
    const vec3 EYE = v3(0, 0, 4.0);                                              // c1:20
    memset(image, 0, sizeof(double) * (size_t)width * height * 3);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < height; ++i) {
        chap1::Tracer tr;
        tr.shapes = shapes;
        tr.gen.seed(seed * 2654435761u + (unsigned)i * 40503u + 12345u);
        for (int k = 0; k < samples; ++k) {
            double* p = image + (size_t)i * width * 3;
            for (int j = 0; j < width; ++j) {  // c1:414-475
                double x = 2.0 * double(j) / double(width) - 1.0;
                double y = 2.0 * double(height - i) / double(height) - 1.0;
                x += (tr.randf() - 0.5f) / double(width);
                y += (tr.randf() - 0.5f) / double(height);
                const vec3 coord = v3((float)x, (float)y, (float)SCREEN_Z);
                const vec3 direction = normalize(coord - EYE);
                chap1::Ray ray;
                ray.startpoint = coord, ray.direction = direction;
                const chap1::HitResult res = tr.shoot(ray);
                vec3 color = v3(0, 0, 0);
                if (res.isHit) {
                    if (res.material.isEmissive) {
                        color = res.material.color;  // not scaled by BRIGHTNESS in the reference (c1:436-439)
                    } else {
                        chap1::Ray randomRay;
                        randomRay.startpoint = res.HitPoint;
                        randomRay.direction = tr.randomDirection(res.material.normal);
                        color = tr.scatter(ray, res, randomRay, 0, 1.0f, true);
                        color = v3((float)(color.x * BRIGHTNESS), (float)(color.y * BRIGHTNESS), (float)(color.z * BRIGHTNESS));  // c1:469
                    }
                }
                *p += color.x, ++p;
                *p += color.y, ++p;
                *p += color.z, ++p;
            }
        }
    }
    return RT_OK;
}

// imshow(), c1:326-347: gamma 1/1.9, *255, clamp, truncate
extern "C" int rt_chap1_to_rgb8(const double* image, int width, int height, unsigned char* out) {
    if (!image || !out || width <= 0 || height <= 0) return set_error(RT_ERR_INVALID_ARG, "rt_chap1_to_rgb8: bad arguments");
    const size_t n = (size_t)width * height * 3;
    for (size_t k = 0; k < n; ++k) {
        double v = pow(image[k], 1.0f / 1.9f) * 255;
        v = (v >= 0.0) ? ((v <= 255.0) ? v : 255.0) : 0.0;
        out[k] = (unsigned char)v;
    }
    return RT_OK;
}
