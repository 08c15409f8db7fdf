// ORACLE — test infrastructure, NOT product code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load this library; the product (ray_tracing_amd/) never does.
//
// A plain CPU restatement (fp32, scalar, line by line) of the reference's algorithm for the path-loop hot path and of
// the host code that feeds it.  File abbreviations as in SURVEY.md: c2 = path_tracing_2chap/path_tracing_2.cpp,
// c5 = path_tracing_5chap/path_tracing_5chap.cpp, f3/f4/f5 = path_tracing_{3,4,5}chap/shaders/fshader.fsh,
// hdr = path_tracing_5chap/lib/hdrloader.cpp.
//
// Third-party arithmetic the reference leans on and that is absent from /root/reference (README.md:70, no lockfile):
//   * GLM >= 0.9.9.5 (vec3 ops, normalize, cross, dot, min/max, lookAt, inverse, rotate/translate/scale): restated here
//     from its published scalar formulas;
//   * libstdc++ std::sort / std::lower_bound: used directly (this container's g++ 11.4);
//   * GLSL built-ins: mix, clamp, reflect, min/max, NEAREST texture2D restated; sin/cos/atan/asin/log/pow are
//     driver-defined in the reference -> either libm (use_libm=1) or the build's fixed fp32 sequences (oracle_math.h).
// PARITY PINNING: the reference ships no tests or golden vectors and cannot be built here without a GLM stand-in.  The
// oracle is pinned by the values SURVEY.md §8c records from the reference source run in this container (triangle/node
// counts, the c2:605-609 test ray -> triangle 2054 / 0.76238, 45 494 primary hits + hit-map hash, cache[23]).  The
// BRDF/MIS loop itself has no independent reference output: for it, parity is "restatement vs kernel" — unpinned.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fopenmp -shared -fPIC oracle.cpp -o liboracle.so   (see Makefile)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "oracle_math.h"

namespace orc {

// ---------------------------------------------------------------------------------------------------------------
// GLM-shaped vec3 (scalar code path of GLM: component-wise ops, dot = (x+y)+z, normalize = v * inversesqrt(dot))
// ---------------------------------------------------------------------------------------------------------------
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
};
static inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator/(const vec3& a, const vec3& b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, const vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
static inline vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec3& operator+=(vec3& a, const vec3& b) { a = a + b; return a; }
static inline vec3& operator*=(vec3& a, const vec3& b) { a = a * b; return a; }
static inline float dot(const vec3& a, const vec3& b) {
    const vec3 tmp = a * b;
    return tmp.x + tmp.y + tmp.z;
}
static inline vec3 cross(const vec3& x, const vec3& y) {
    return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline vec3 normalize(const vec3& v) { return v * (1.0f / sqrtf(dot(v, v))); }
static inline float min(float x, float y) { return (y < x) ? y : x; }  // glm::min / GLSL min
static inline float max(float x, float y) { return (x < y) ? y : x; }  // glm::max / GLSL max
static inline vec3 min(const vec3& a, const vec3& b) { return vec3(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
static inline vec3 max(const vec3& a, const vec3& b) { return vec3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
static inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
static inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
static inline vec3 mix(const vec3& x, const vec3& y, float a) { return x * (1.0f - a) + y * a; }
static inline vec3 reflect(const vec3& I, const vec3& N) { return I - N * dot(N, I) * 2.0f; }
static inline float radians(float d) { return d * 0.01745329251994329576923690768489f; }

struct mat4 {
    float m[4][4];  // m[col][row]
};
static mat4 identity() {
    mat4 r;
    memset(&r, 0, sizeof(r));
    r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
    return r;
}
static mat4 mul(const mat4& a, const mat4& b) {  // glm operator*(mat4, mat4)
    mat4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.m[j][i] = a.m[0][i] * b.m[j][0] + a.m[1][i] * b.m[j][1] + a.m[2][i] * b.m[j][2] + a.m[3][i] * b.m[j][3];
    return r;
}
static void mulv(const mat4& a, const float v[4], float o[4]) {  // glm operator*(mat4, vec4)
    for (int i = 0; i < 4; ++i) {
        const float Add0 = a.m[0][i] * v[0] + a.m[1][i] * v[1];
        const float Add1 = a.m[2][i] * v[2] + a.m[3][i] * v[3];
        o[i] = Add0 + Add1;
    }
}
static mat4 rotate(const mat4& m, float angle, const vec3& v) {  // glm::rotate
    const float c = cosf(angle), s = sinf(angle);
    const vec3 axis = normalize(v);
    const vec3 temp = (1.0f - c) * axis;
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = temp.x * axis.y + s * axis.z;
    R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;
    R[2][1] = temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    mat4 r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 4; ++i) r.m[j][i] = m.m[0][i] * R[j][0] + m.m[1][i] * R[j][1] + m.m[2][i] * R[j][2];
    for (int i = 0; i < 4; ++i) r.m[3][i] = m.m[3][i];
    return r;
}
static mat4 scale(const mat4& m, const vec3& v) {
    mat4 r;
    for (int i = 0; i < 4; ++i) r.m[0][i] = m.m[0][i] * v.x, r.m[1][i] = m.m[1][i] * v.y, r.m[2][i] = m.m[2][i] * v.z, r.m[3][i] = m.m[3][i];
    return r;
}
static mat4 translate(const mat4& m, const vec3& v) {
    mat4 r = m;
    for (int i = 0; i < 4; ++i) r.m[3][i] = m.m[0][i] * v.x + m.m[1][i] * v.y + m.m[2][i] * v.z + m.m[3][i];
    return r;
}
static mat4 lookAt(const vec3& eye, const vec3& center, const vec3& up) {  // glm::lookAtRH
    const vec3 f(normalize(center - eye));
    const vec3 s(normalize(cross(f, up)));
    const vec3 u(cross(s, f));
    mat4 r = identity();
    r.m[0][0] = s.x, r.m[1][0] = s.y, r.m[2][0] = s.z;
    r.m[0][1] = u.x, r.m[1][1] = u.y, r.m[2][1] = u.z;
    r.m[0][2] = -f.x, r.m[1][2] = -f.y, r.m[2][2] = -f.z;
    r.m[3][0] = -dot(s, eye), r.m[3][1] = -dot(u, eye), r.m[3][2] = dot(f, eye);
    return r;
}
static mat4 inverse(const mat4& M) {  // glm compute_inverse<4,4,float>
    const float(*m)[4] = M.m;
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3], Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3], Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2], Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3], Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2], Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1], Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    float Fac0[4] = {Coef00, Coef00, Coef02, Coef03}, Fac1[4] = {Coef04, Coef04, Coef06, Coef07}, Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    float Fac3[4] = {Coef12, Coef12, Coef14, Coef15}, Fac4[4] = {Coef16, Coef16, Coef18, Coef19}, Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    float Vec0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]}, Vec1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
    float Vec2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]}, Vec3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
    mat4 Inv;
    for (int i = 0; i < 4; ++i) {
        const float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
        Inv.m[0][i] = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i]) * sa;
        Inv.m[1][i] = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i]) * sb;
        Inv.m[2][i] = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i]) * sa;
        Inv.m[3][i] = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i]) * sb;
    }
    const float Dot0[4] = {m[0][0] * Inv.m[0][0], m[0][1] * Inv.m[1][0], m[0][2] * Inv.m[2][0], m[0][3] * Inv.m[3][0]};
    const float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    const float OneOverDeterminant = 1.0f / Dot1;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) Inv.m[j][i] *= OneOverDeterminant;
    return Inv;
}

// ---------------------------------------------------------------------------------------------------------------
// Host scene code, c5:23-70, 155-169, 249-397, 462-698; c2:142-183
// ---------------------------------------------------------------------------------------------------------------
#define INF 114514.0

struct Material {  // c5:23-39
    vec3 emissive = vec3(0, 0, 0);
    vec3 baseColor = vec3(1, 1, 1);
    float subsurface = 0.0;
    float metallic = 0.0;
    float specular = 0.0;
    float specularTint = 0.0;
    float roughness = 0.0;
    float anisotropic = 0.0;
    float sheen = 0.0;
    float sheenTint = 0.0;
    float clearcoat = 0.0;
    float clearcoatGloss = 0.0;
    float IOR = 1.0;
    float transmission = 0.0;
};
struct Triangle {  // c5:41-46
    vec3 p1, p2, p3;
    vec3 n1, n2, n3;
    Material material;
};
struct BVHNode {  // c5:48-53
    int left, right;
    vec3 AA, BB;
    int n, index;
};

static bool cmpx(const Triangle& t1, const Triangle& t2) {  // c5:155-159
    vec3 center1 = (t1.p1 + t1.p2 + t1.p3) / vec3(3, 3, 3);
    vec3 center2 = (t2.p1 + t2.p2 + t2.p3) / vec3(3, 3, 3);
    return center1.x < center2.x;
}
static bool cmpy(const Triangle& t1, const Triangle& t2) {
    vec3 center1 = (t1.p1 + t1.p2 + t1.p3) / vec3(3, 3, 3);
    vec3 center2 = (t2.p1 + t2.p2 + t2.p3) / vec3(3, 3, 3);
    return center1.y < center2.y;
}
static bool cmpz(const Triangle& t1, const Triangle& t2) {
    vec3 center1 = (t1.p1 + t1.p2 + t1.p3) / vec3(3, 3, 3);
    vec3 center2 = (t2.p1 + t2.p2 + t2.p3) / vec3(3, 3, 3);
    return center1.z < center2.z;
}

static mat4 getTransformMatrix(vec3 rotateCtrl, vec3 translateCtrl, vec3 scaleCtrl) {  // c5:249-265
    mat4 unit = identity();
    mat4 scl = scale(unit, scaleCtrl);
    mat4 trn = translate(unit, translateCtrl);
    mat4 rot = unit;
    rot = rotate(rot, radians(rotateCtrl.x), vec3(1, 0, 0));
    rot = rotate(rot, radians(rotateCtrl.y), vec3(0, 1, 0));
    rot = rotate(rot, radians(rotateCtrl.z), vec3(0, 0, 1));
    return mul(mul(trn, rot), scl);
}

static int readObj(std::string filepath, std::vector<Triangle>& triangles, Material material, mat4 trans, bool smoothNormal) {  // c5:267-397
    std::vector<vec3> vertices;
    std::vector<unsigned> indices;
    std::ifstream fin(filepath);
    std::string line;
    if (!fin.is_open()) return -1;
    float maxx = -11451419.19, maxy = -11451419.19, maxz = -11451419.19;
    float minx = 11451419.19, miny = 11451419.19, minz = 11451419.19;
    while (std::getline(fin, line)) {
        std::istringstream sin(line);
        std::string type;
        float x, y, z;
        int v0, v1, v2, vn0, vn1, vn2, vt0, vt1, vt2;
        char slash;
        int slashcnt = 0;
        for (size_t i = 0; i < line.size(); ++i)
            if (line[i] == '/') ++slashcnt;
        sin >> type;
        if (type == "v") {
            sin >> x >> y >> z;
            vertices.push_back(vec3(x, y, z));
            maxx = max(maxx, x), maxy = max(maxy, y), maxz = max(maxz, z);
            minx = min(minx, x), miny = min(miny, y), minz = min(minz, z);
        }
        if (type == "f") {
            if (slashcnt == 6) {
                sin >> v0 >> slash >> vt0 >> slash >> vn0;
                sin >> v1 >> slash >> vt1 >> slash >> vn1;
                sin >> v2 >> slash >> vt2 >> slash >> vn2;
            } else if (slashcnt == 3) {
                sin >> v0 >> slash >> vt0;
                sin >> v1 >> slash >> vt1;
                sin >> v2 >> slash >> vt2;
            } else {
                sin >> v0 >> v1 >> v2;
            }
            indices.push_back(v0 - 1);
            indices.push_back(v1 - 1);
            indices.push_back(v2 - 1);
        }
    }
    float lenx = maxx - minx, leny = maxy - miny, lenz = maxz - minz;
    float maxaxis = max(lenx, max(leny, lenz));
    for (auto& v : vertices) {
        v.x /= maxaxis;
        v.y /= maxaxis;
        v.z /= maxaxis;
    }
    for (auto& v : vertices) {
        float vv[4] = {v.x, v.y, v.z, 1}, o[4];
        mulv(trans, vv, o);
        v = vec3(o[0], o[1], o[2]);
    }
    std::vector<vec3> normals(vertices.size(), vec3(0, 0, 0));
    for (size_t i = 0; i < indices.size(); i += 3) {
        vec3 p1 = vertices[indices[i]], p2 = vertices[indices[i + 1]], p3 = vertices[indices[i + 2]];
        vec3 n = normalize(cross(p2 - p1, p3 - p1));
        normals[indices[i]] += n;
        normals[indices[i + 1]] += n;
        normals[indices[i + 2]] += n;
    }
    int offset = triangles.size();
    triangles.resize(offset + indices.size() / 3);
    for (size_t i = 0; i < indices.size(); i += 3) {
        Triangle& t = triangles[offset + i / 3];
        t.p1 = vertices[indices[i]];
        t.p2 = vertices[indices[i + 1]];
        t.p3 = vertices[indices[i + 2]];
        if (!smoothNormal) {
            vec3 n = normalize(cross(t.p2 - t.p1, t.p3 - t.p1));
            t.n1 = n, t.n2 = n, t.n3 = n;
        } else {
            t.n1 = normalize(normals[indices[i]]);
            t.n2 = normalize(normals[indices[i + 1]]);
            t.n3 = normalize(normals[indices[i + 2]]);
        }
        t.material = material;
    }
    return 0;
}

// c2:142-183 + the edit of c2:590-594; chapter-2 triangles carry positions only (c2:38-47): normals/material are
// filled with the flat normal and the default material so that the record has the chapter 3-5 layout.
static int readObjChap2(std::string filepath, std::vector<Triangle>& triangles, float scl, float yoff) {
    std::vector<vec3> vertices;
    std::vector<unsigned> indices;
    std::ifstream fin(filepath);
    std::string line;
    if (!fin.is_open()) return -1;
    while (getline(fin, line)) {
        std::istringstream sin(line);
        std::string type;
        float x, y, z;
        int v0, v1, v2;
        sin >> type;
        if (type == "v") {
            sin >> x >> y >> z;
            vertices.push_back(vec3(x, y, z));
        }
        if (type == "f") {
            sin >> v0 >> v1 >> v2;
            indices.push_back(v0 - 1);
            indices.push_back(v1 - 1);
            indices.push_back(v2 - 1);
        }
    }
    for (auto& v : vertices) {
        v.x *= scl, v.y *= scl, v.z *= scl;
        v.y -= yoff;
    }
    for (size_t i = 0; i < indices.size(); i += 3) {
        Triangle t;
        t.p1 = vertices[indices[i]], t.p2 = vertices[indices[i + 1]], t.p3 = vertices[indices[i + 2]];
        vec3 n = normalize(cross(t.p2 - t.p1, t.p3 - t.p1));
        t.n1 = n, t.n2 = n, t.n3 = n;
        triangles.push_back(t);
    }
    return 0;
}

static int buildBVHwithSAH(std::vector<Triangle>& triangles, std::vector<BVHNode>& nodes, int l, int r, int n) {  // c5:462-593
    if (l > r) return 0;
    nodes.push_back(BVHNode());
    int id = nodes.size() - 1;
    nodes[id].left = nodes[id].right = nodes[id].n = nodes[id].index = 0;
    nodes[id].AA = vec3(1145141919, 1145141919, 1145141919);
    nodes[id].BB = vec3(-1145141919, -1145141919, -1145141919);
    for (int i = l; i <= r; i++) {
        float minx = min(triangles[i].p1.x, min(triangles[i].p2.x, triangles[i].p3.x));
        float miny = min(triangles[i].p1.y, min(triangles[i].p2.y, triangles[i].p3.y));
        float minz = min(triangles[i].p1.z, min(triangles[i].p2.z, triangles[i].p3.z));
        nodes[id].AA.x = min(nodes[id].AA.x, minx);
        nodes[id].AA.y = min(nodes[id].AA.y, miny);
        nodes[id].AA.z = min(nodes[id].AA.z, minz);
        float maxx = max(triangles[i].p1.x, max(triangles[i].p2.x, triangles[i].p3.x));
        float maxy = max(triangles[i].p1.y, max(triangles[i].p2.y, triangles[i].p3.y));
        float maxz = max(triangles[i].p1.z, max(triangles[i].p2.z, triangles[i].p3.z));
        nodes[id].BB.x = max(nodes[id].BB.x, maxx);
        nodes[id].BB.y = max(nodes[id].BB.y, maxy);
        nodes[id].BB.z = max(nodes[id].BB.z, maxz);
    }
    if ((r - l + 1) <= n) {
        nodes[id].n = r - l + 1;
        nodes[id].index = l;
        return id;
    }
    float Cost = INF;
    int Axis = 0;
    int Split = (l + r) / 2;
    for (int axis = 0; axis < 3; axis++) {
        if (axis == 0) std::sort(&triangles[0] + l, &triangles[0] + r + 1, cmpx);
        if (axis == 1) std::sort(&triangles[0] + l, &triangles[0] + r + 1, cmpy);
        if (axis == 2) std::sort(&triangles[0] + l, &triangles[0] + r + 1, cmpz);
        std::vector<vec3> leftMax(r - l + 1, vec3(-INF, -INF, -INF));
        std::vector<vec3> leftMin(r - l + 1, vec3(INF, INF, INF));
        for (int i = l; i <= r; i++) {
            Triangle& t = triangles[i];
            int bias = (i == l) ? 0 : 1;
            leftMax[i - l].x = max(leftMax[i - l - bias].x, max(t.p1.x, max(t.p2.x, t.p3.x)));
            leftMax[i - l].y = max(leftMax[i - l - bias].y, max(t.p1.y, max(t.p2.y, t.p3.y)));
            leftMax[i - l].z = max(leftMax[i - l - bias].z, max(t.p1.z, max(t.p2.z, t.p3.z)));
            leftMin[i - l].x = min(leftMin[i - l - bias].x, min(t.p1.x, min(t.p2.x, t.p3.x)));
            leftMin[i - l].y = min(leftMin[i - l - bias].y, min(t.p1.y, min(t.p2.y, t.p3.y)));
            leftMin[i - l].z = min(leftMin[i - l - bias].z, min(t.p1.z, min(t.p2.z, t.p3.z)));
        }
        std::vector<vec3> rightMax(r - l + 1, vec3(-INF, -INF, -INF));
        std::vector<vec3> rightMin(r - l + 1, vec3(INF, INF, INF));
        for (int i = r; i >= l; i--) {
            Triangle& t = triangles[i];
            int bias = (i == r) ? 0 : 1;
            rightMax[i - l].x = max(rightMax[i - l + bias].x, max(t.p1.x, max(t.p2.x, t.p3.x)));
            rightMax[i - l].y = max(rightMax[i - l + bias].y, max(t.p1.y, max(t.p2.y, t.p3.y)));
            rightMax[i - l].z = max(rightMax[i - l + bias].z, max(t.p1.z, max(t.p2.z, t.p3.z)));
            rightMin[i - l].x = min(rightMin[i - l + bias].x, min(t.p1.x, min(t.p2.x, t.p3.x)));
            rightMin[i - l].y = min(rightMin[i - l + bias].y, min(t.p1.y, min(t.p2.y, t.p3.y)));
            rightMin[i - l].z = min(rightMin[i - l + bias].z, min(t.p1.z, min(t.p2.z, t.p3.z)));
        }
        float cost = INF;
        float split = l;
        for (int i = l; i <= r - 1; ++i) {
            float lenx, leny, lenz;
            vec3 leftAA = leftMin[i - l];
            vec3 leftBB = leftMax[i - l];
            lenx = leftBB.x - leftAA.x;
            leny = leftBB.y - leftAA.y;
            lenz = leftBB.z - leftAA.z;
            float leftS = 2.0 * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
            float leftcost = leftS * (i - l + 1);
            vec3 rightAA = rightMin[i - l + 1];
            vec3 rightBB = rightMax[i - l + 1];
            lenx = rightBB.x - rightAA.x;
            leny = rightBB.y - rightAA.y;
            lenz = rightBB.z - rightAA.z;
            float rightS = 2.0 * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
            float rightcost = rightS * (r - i);
            float totalCost = rightcost + leftcost;
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
    if (Axis == 0) std::sort(triangles.begin() + l, triangles.begin() + r + 1, cmpx);
    if (Axis == 1) std::sort(triangles.begin() + l, triangles.begin() + r + 1, cmpy);
    if (Axis == 2) std::sort(triangles.begin() + l, triangles.begin() + r + 1, cmpz);
    int left = buildBVHwithSAH(triangles, nodes, l, Split, n);  // C++17: RHS sequenced before nodes[id] (c5:587)
    nodes[id].left = left;
    int right = buildBVHwithSAH(triangles, nodes, Split + 1, r, n);
    nodes[id].right = right;
    return id;
}

static int buildBVH(std::vector<Triangle>& triangles, std::vector<BVHNode>& nodes, int l, int r, int n) {  // c5:399-460
    if (l > r) return 0;
    nodes.push_back(BVHNode());
    int id = nodes.size() - 1;
    nodes[id].left = nodes[id].right = nodes[id].n = nodes[id].index = 0;
    nodes[id].AA = vec3(1145141919, 1145141919, 1145141919);
    nodes[id].BB = vec3(-1145141919, -1145141919, -1145141919);
    for (int i = l; i <= r; i++) {
        float minx = min(triangles[i].p1.x, min(triangles[i].p2.x, triangles[i].p3.x));
        float miny = min(triangles[i].p1.y, min(triangles[i].p2.y, triangles[i].p3.y));
        float minz = min(triangles[i].p1.z, min(triangles[i].p2.z, triangles[i].p3.z));
        nodes[id].AA.x = min(nodes[id].AA.x, minx), nodes[id].AA.y = min(nodes[id].AA.y, miny), nodes[id].AA.z = min(nodes[id].AA.z, minz);
        float maxx = max(triangles[i].p1.x, max(triangles[i].p2.x, triangles[i].p3.x));
        float maxy = max(triangles[i].p1.y, max(triangles[i].p2.y, triangles[i].p3.y));
        float maxz = max(triangles[i].p1.z, max(triangles[i].p2.z, triangles[i].p3.z));
        nodes[id].BB.x = max(nodes[id].BB.x, maxx), nodes[id].BB.y = max(nodes[id].BB.y, maxy), nodes[id].BB.z = max(nodes[id].BB.z, maxz);
    }
    if ((r - l + 1) <= n) {
        nodes[id].n = r - l + 1;
        nodes[id].index = l;
        return id;
    }
    float lenx = nodes[id].BB.x - nodes[id].AA.x, leny = nodes[id].BB.y - nodes[id].AA.y, lenz = nodes[id].BB.z - nodes[id].AA.z;
    if (lenx >= leny && lenx >= lenz) std::sort(triangles.begin() + l, triangles.begin() + r + 1, cmpx);
    if (leny >= lenx && leny >= lenz) std::sort(triangles.begin() + l, triangles.begin() + r + 1, cmpy);
    if (lenz >= lenx && lenz >= leny) std::sort(triangles.begin() + l, triangles.begin() + r + 1, cmpz);
    int mid = (l + r) / 2;
    int left = buildBVH(triangles, nodes, l, mid, n);
    int right = buildBVH(triangles, nodes, mid + 1, r, n);
    nodes[id].left = left;
    nodes[id].right = right;
    return id;
}

static float* calculateHdrCache(const float* HDR, int width, int height) {  // c5:595-698
    float lumSum = 0.0;
    std::vector<std::vector<float>> pdf(height);
    for (auto& line : pdf) line.resize(width);
    for (int i = 0; i < height; i++) {
        for (int j = 0; j < width; j++) {
            float R = HDR[3 * (i * width + j)];
            float G = HDR[3 * (i * width + j) + 1];
            float B = HDR[3 * (i * width + j) + 2];
            float lum = 0.2 * R + 0.7 * G + 0.1 * B;
            pdf[i][j] = lum;
            lumSum += lum;
        }
    }
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j) pdf[i][j] /= lumSum;
    std::vector<float> pdf_x_margin;
    pdf_x_margin.resize(width);
    for (int j = 0; j < width; j++)
        for (int i = 0; i < height; i++) pdf_x_margin[j] += pdf[i][j];
    std::vector<float> cdf_x_margin = pdf_x_margin;
    for (int i = 1; i < width; i++) cdf_x_margin[i] += cdf_x_margin[i - 1];
    std::vector<std::vector<float>> pdf_y_condiciton = pdf;
    for (int j = 0; j < width; j++)
        for (int i = 0; i < height; i++) pdf_y_condiciton[i][j] /= pdf_x_margin[j];
    std::vector<std::vector<float>> cdf_y_condiciton = pdf_y_condiciton;
    for (int j = 0; j < width; j++)
        for (int i = 1; i < height; i++) cdf_y_condiciton[i][j] += cdf_y_condiciton[i - 1][j];
    std::vector<std::vector<float>> temp = cdf_y_condiciton;
    cdf_y_condiciton = std::vector<std::vector<float>>(width);
    for (auto& line : cdf_y_condiciton) line.resize(height);
    for (int j = 0; j < width; j++)
        for (int i = 0; i < height; i++) cdf_y_condiciton[j][i] = temp[i][j];
    float* cache = new float[width * height * 3];
    for (int j = 0; j < width; ++j) {
        for (int i = 0; i < height; ++i) {
            float xi_1 = float(i) / height;
            float xi_2 = float(j) / width;
            int x = std::lower_bound(cdf_x_margin.begin(), cdf_x_margin.end(), xi_1) - cdf_x_margin.begin();
            int xc = x < width ? x : width - 1;  // the reference indexes [width] here when xi_1 > cdf.back() (UB)
            int y = std::lower_bound(cdf_y_condiciton[xc].begin(), cdf_y_condiciton[xc].end(), xi_2) - cdf_y_condiciton[xc].begin();
            cache[3 * (i * width + j)] = float(x) / width;
            cache[3 * (i * width + j) + 1] = float(y) / height;
            cache[3 * (i * width + j) + 2] = pdf[i][j];
        }
    }
    return cache;
}

// hdr:29-191 with %d instead of %ld (hdr:68 writes two longs into two ints on LP64)
typedef unsigned char RGBE[4];
static bool oldDecrunch(RGBE* scanline, int len, FILE* file) {
    int i, rshift = 0;
    while (len > 0) {
        scanline[0][0] = fgetc(file), scanline[0][1] = fgetc(file), scanline[0][2] = fgetc(file), scanline[0][3] = fgetc(file);
        if (feof(file)) return false;
        if (scanline[0][0] == 1 && scanline[0][1] == 1 && scanline[0][2] == 1) {
            for (i = scanline[0][3] << rshift; i > 0 && len > 0; i--) {
                memcpy(&scanline[0][0], &scanline[-1][0], 4);
                scanline++, len--;
            }
            rshift += 8;
        } else {
            scanline++, len--;
            rshift = 0;
        }
    }
    return true;
}
static bool decrunch(RGBE* scanline, int len, FILE* file) {
    int i, j;
    if (len < 8 || len > 0x7fff) return oldDecrunch(scanline, len, file);
    i = fgetc(file);
    if (i != 2) {
        fseek(file, -1, SEEK_CUR);
        return oldDecrunch(scanline, len, file);
    }
    scanline[0][1] = fgetc(file);
    scanline[0][2] = fgetc(file);
    i = fgetc(file);
    if (scanline[0][1] != 2 || scanline[0][2] & 128) {
        scanline[0][0] = 2;
        scanline[0][3] = i;
        return oldDecrunch(scanline + 1, len - 1, file);
    }
    for (i = 0; i < 4; i++) {
        for (j = 0; j < len;) {
            unsigned char code = fgetc(file);
            if (feof(file)) return false;
            if (code > 128) {
                code &= 127;
                unsigned char val = fgetc(file);
                while (code-- && j < len) scanline[j++][i] = val;
            } else {
                while (code-- && j < len) scanline[j++][i] = fgetc(file);
            }
        }
    }
    return feof(file) ? false : true;
}
static float convertComponent(int expo, int val) {
    float v = val / 256.0f;
    float d = (float)pow(2, expo);
    return v * d;
}

// ---------------------------------------------------------------------------------------------------------------
// The fragment shader, restated.  Math policy: driver-defined built-ins.
// ---------------------------------------------------------------------------------------------------------------
struct LibmMath {
    static float sin_(float x) { return sinf(x); }
    static float cos_(float x) { return cosf(x); }
    static float atan2_(float y, float x) { return atan2f(y, x); }
    static float asin_(float x) { return asinf(x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x)); }
    static float log_(float x) { return logf(x); }
    static float pow_(float x, float y) { return powf(x, y); }
};
struct DetMath {
    static float sin_(float x) { return det_sin(x); }
    static float cos_(float x) { return det_cos(x); }
    static float atan2_(float y, float x) { return det_atan2(y, x); }
    static float asin_(float x) { return det_asin(x); }
    static float log_(float x) { return det_log(x); }
    static float pow_(float x, float y) { return det_pow(x, y); }
};

#undef INF
static const float PI = 3.1415926f;   // f5:27
static const float INF = 114514.0f;   // f5:28
#define SIZE_TRIANGLE 12
#define SIZE_BVHNODE 4

struct GTriangle {
    vec3 p1, p2, p3;
    vec3 n1, n2, n3;
};
struct GBVHNode {
    int left, right, n, index;
    vec3 AA, BB;
};
struct Ray {
    vec3 startPoint, direction;
};
struct HitResult {
    bool isHit = false;
    bool isInside = false;
    float distance = 0;
    vec3 hitPoint, normal, viewDir;
    Material material;
    int triIndex = -1;  // bookkeeping only (not in the shader)
};
struct Counters {
    uint64_t paths = 0, rays = 0, internal = 0, leaves = 0, tris = 0, updates = 0;
};

static const uint32_t V[8 * 32] = {  // f5:339
    2147483648u, 1073741824u, 536870912u, 268435456u, 134217728u, 67108864u, 33554432u, 16777216u, 8388608u, 4194304u, 2097152u, 1048576u, 524288u, 262144u, 131072u, 65536u, 32768u, 16384u, 8192u, 4096u, 2048u, 1024u, 512u, 256u, 128u, 64u, 32u, 16u, 8u, 4u, 2u, 1u, 2147483648u, 3221225472u, 2684354560u, 4026531840u, 2281701376u, 3422552064u, 2852126720u, 4278190080u, 2155872256u, 3233808384u, 2694840320u, 4042260480u, 2290614272u, 3435921408u, 2863267840u, 4294901760u, 2147516416u, 3221274624u, 2684395520u, 4026593280u, 2281736192u, 3422604288u, 2852170240u, 4278255360u, 2155905152u, 3233857728u, 2694881440u, 4042322160u, 2290649224u, 3435973836u, 2863311530u, 4294967295u, 2147483648u, 3221225472u, 1610612736u, 2415919104u, 3892314112u, 1543503872u, 2382364672u, 3305111552u, 1753219072u, 2629828608u, 3999268864u, 1435500544u, 2154299392u, 3231449088u, 1626210304u, 2421489664u, 3900735488u, 1556135936u, 2388680704u, 3314585600u, 1751705600u, 2627492864u, 4008611328u, 1431684352u, 2147543168u, 3221249216u, 1610649184u, 2415969680u, 3892340840u, 1543543964u, 2382425838u, 3305133397u, 2147483648u, 3221225472u, 536870912u, 1342177280u, 4160749568u, 1946157056u, 2717908992u, 2466250752u, 3632267264u, 624951296u, 1507852288u, 3872391168u, 2013790208u, 3020685312u, 2181169152u, 3271884800u, 546275328u, 1363623936u, 4226424832u, 1977167872u, 2693105664u, 2437829632u, 3689389568u, 635137280u, 1484783744u, 3846176960u, 2044723232u, 3067084880u, 2148008184u, 3222012020u, 537002146u, 1342505107u, 2147483648u, 1073741824u, 536870912u, 2952790016u, 4160749568u, 3690987520u, 2046820352u, 2634022912u, 1518338048u, 801112064u, 2707423232u, 4038066176u, 3666345984u, 1875116032u, 2170683392u, 1085997056u, 579305472u, 3016343552u, 4217741312u, 3719483392u, 2013407232u, 2617981952u, 1510979072u, 755882752u, 2726789248u, 4090085440u, 3680870432u, 1840435376u, 2147625208u, 1074478300u, 537900666u, 2953698205u, 2147483648u, 1073741824u, 1610612736u, 805306368u, 2818572288u, 335544320u, 2113929216u, 3472883712u, 2290089984u, 3829399552u, 3059744768u, 1127219200u, 3089629184u, 4199809024u, 3567124480u, 1891565568u, 394297344u, 3988799488u, 920674304u, 4193267712u, 2950604800u, 3977188352u, 3250028032u, 129093376u, 2231568512u, 2963678272u, 4281226848u, 432124720u, 803643432u, 1633613396u, 2672665246u, 3170194367u, 2147483648u, 3221225472u, 2684354560u, 3489660928u, 1476395008u, 2483027968u, 1040187392u, 3808428032u, 3196059648u, 599785472u, 505413632u, 4077912064u, 1182269440u, 1736704000u, 2017853440u, 2221342720u, 3329785856u, 2810494976u, 3628507136u, 1416089600u, 2658719744u, 864310272u, 3863387648u, 3076993792u, 553150080u, 272922560u, 4167467040u, 1148698640u, 1719673080u, 2009075780u, 2149644390u, 3222291575u, 2147483648u, 1073741824u, 2684354560u, 1342177280u, 2281701376u, 1946157056u, 436207616u, 2566914048u, 2625634304u, 3208642560u, 2720006144u, 2098200576u, 111673344u, 2354315264u, 3464626176u, 4027383808u, 2886631424u, 3770826752u, 1691164672u, 3357462528u, 1993345024u, 3752330240u, 873073152u, 2870150400u, 1700563072u, 87021376u, 1097028000u, 1222351248u, 1560027592u, 2977959924u, 23268898u, 437609937u};

template <class M, int CHAP>
struct Shader {
    // uniforms, f5:8-23
    int frameCounter, nTriangles, nNodes, width, height, hdrResolution;
    const float* triangles;  // samplerBuffer, RGB32F texels
    const float* nodes;
    const float* hdrMap;     // sampler2D RGB32F, hdrW x hdrH, NEAREST, CLAMP_TO_EDGE
    const float* hdrCache;
    int hdrW, hdrH;
    vec3 eye;
    mat4 cameraRotate;
    // varying
    float pixx, pixy;
    // global
    uint32_t seed;
    Counters* cnt;

    vec3 texelFetch(const float* buf, int i) const { return vec3(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]); }
    vec3 texture2D(const float* tex, float u, float v) const {
        if (!tex) return vec3(0);
        float fx = floorf(u * (float)hdrW), fy = floorf(v * (float)hdrH);
        int ix = (fx >= 0.0f) ? ((fx <= (float)(hdrW - 1)) ? (int)fx : hdrW - 1) : 0;
        int iy = (fy >= 0.0f) ? ((fy <= (float)(hdrH - 1)) ? (int)fy : hdrH - 1) : 0;
        return texelFetch(tex, iy * hdrW + ix);
    }

    GTriangle getTriangle(int i) const {  // f5:101-115
        int offset = i * SIZE_TRIANGLE;
        GTriangle t;
        t.p1 = texelFetch(triangles, offset + 0);
        t.p2 = texelFetch(triangles, offset + 1);
        t.p3 = texelFetch(triangles, offset + 2);
        t.n1 = texelFetch(triangles, offset + 3);
        t.n2 = texelFetch(triangles, offset + 4);
        t.n3 = texelFetch(triangles, offset + 5);
        return t;
    }
    Material getMaterial(int i) const {  // f5:118-143
        Material m;
        int offset = i * SIZE_TRIANGLE;
        vec3 param1 = texelFetch(triangles, offset + 8);
        vec3 param2 = texelFetch(triangles, offset + 9);
        vec3 param3 = texelFetch(triangles, offset + 10);
        vec3 param4 = texelFetch(triangles, offset + 11);
        m.emissive = texelFetch(triangles, offset + 6);
        m.baseColor = texelFetch(triangles, offset + 7);
        m.subsurface = param1.x, m.metallic = param1.y, m.specular = param1.z;
        m.specularTint = param2.x, m.roughness = param2.y, m.anisotropic = param2.z;
        m.sheen = param3.x, m.sheenTint = param3.y, m.clearcoat = param3.z;
        m.clearcoatGloss = param4.x, m.IOR = param4.y, m.transmission = param4.z;
        return m;
    }
    GBVHNode getBVHNode(int i) const {  // f5:146-163
        GBVHNode node;
        int offset = i * SIZE_BVHNODE;
        vec3 childs = texelFetch(nodes, offset + 0);
        vec3 leafInfo = texelFetch(nodes, offset + 1);
        node.left = int(childs.x);
        node.right = int(childs.y);
        node.n = int(leafInfo.x);
        node.index = int(leafInfo.y);
        node.AA = texelFetch(nodes, offset + 2);
        node.BB = texelFetch(nodes, offset + 3);
        return node;
    }

    HitResult hitTriangle(const GTriangle& triangle, const Ray& ray) const {  // f5:168-222 (f4:355-409)
        HitResult res;
        res.distance = INF;
        res.isHit = false;
        res.isInside = false;
        vec3 p1 = triangle.p1, p2 = triangle.p2, p3 = triangle.p3;
        vec3 S = ray.startPoint;
        vec3 d = ray.direction;
        vec3 N = normalize(cross(p2 - p1, p3 - p1));
        if (dot(N, d) > 0.0f) {
            N = -N;
            res.isInside = true;
        }
        if (fabsf(dot(N, d)) < 0.00001f) return res;
        float t = (dot(N, p1) - dot(S, N)) / dot(d, N);
        if (t < 0.0005f) return res;
        vec3 P = S + d * t;
        vec3 c1 = cross(p2 - p1, P - p1);
        vec3 c2 = cross(p3 - p2, P - p2);
        vec3 c3 = cross(p1 - p3, P - p3);
        bool r1 = (dot(c1, N) > 0 && dot(c2, N) > 0 && dot(c3, N) > 0);
        bool r2 = (dot(c1, N) < 0 && dot(c2, N) < 0 && dot(c3, N) < 0);
        if (r1 || r2) {
            res.isHit = true;
            res.hitPoint = P;
            res.distance = t;
            res.normal = N;
            res.viewDir = d;
            float alpha, beta;
            if (CHAP == 5) {  // f5:213-214
                alpha = (-(P.x - p2.x) * (p3.y - p2.y) + (P.y - p2.y) * (p3.x - p2.x)) / (-(p1.x - p2.x) * (p3.y - p2.y) + (p1.y - p2.y) * (p3.x - p2.x) + 1e-7f);
                beta = (-(P.x - p3.x) * (p1.y - p3.y) + (P.y - p3.y) * (p1.x - p3.x)) / (-(p2.x - p3.x) * (p1.y - p3.y) + (p2.y - p3.y) * (p1.x - p3.x) + 1e-7f);
            } else {  // f4:400-401, f3:275-276
                alpha = (-(P.x - p2.x) * (p3.y - p2.y) + (P.y - p2.y) * (p3.x - p2.x)) / (-(p1.x - p2.x - 0.00005f) * (p3.y - p2.y + 0.00005f) + (p1.y - p2.y + 0.00005f) * (p3.x - p2.x + 0.00005f));
                beta = (-(P.x - p3.x) * (p1.y - p3.y) + (P.y - p3.y) * (p1.x - p3.x)) / (-(p2.x - p3.x - 0.00005f) * (p1.y - p3.y + 0.00005f) + (p2.y - p3.y + 0.00005f) * (p1.x - p3.x + 0.00005f));
            }
            float gama = 1.0f - alpha - beta;
            vec3 Nsmooth = alpha * triangle.n1 + beta * triangle.n2 + gama * triangle.n3;
            Nsmooth = normalize(Nsmooth);
            res.normal = (res.isInside) ? (-Nsmooth) : (Nsmooth);
        }
        return res;
    }

    float hitAABB(const Ray& r, vec3 AA, vec3 BB) const {  // f5:225-238
        vec3 invdir = vec3(1.0f) / r.direction;
        vec3 f = (BB - r.startPoint) * invdir;
        vec3 n = (AA - r.startPoint) * invdir;
        vec3 tmax = max(f, n);
        vec3 tmin = min(f, n);
        float t1 = min(tmax.x, min(tmax.y, tmax.z));
        float t0 = max(tmin.x, max(tmin.y, tmin.z));
        return (t1 >= t0) ? ((t0 > 0.0f) ? (t0) : (t1)) : (-1);
    }

    HitResult hitArray(const Ray& ray, int l, int r) const {  // f5:243-256
        HitResult res;
        res.isHit = false;
        res.distance = INF;
        for (int i = l; i <= r; i++) {
            GTriangle triangle = getTriangle(i);
            HitResult rr = hitTriangle(triangle, ray);
            cnt->tris++;
            if (rr.isHit && rr.distance < res.distance) {
                res = rr;
                res.material = getMaterial(i);
                res.triIndex = i;
                cnt->updates++;
            }
        }
        return res;
    }

    HitResult hitBVH(const Ray& ray) const {  // f5:259-311
        HitResult res;
        res.isHit = false;
        res.distance = INF;
        cnt->rays++;
        if (nNodes < 2) return res;
        int stack[256];
        int sp = 0;
        stack[sp++] = 1;
        while (sp > 0) {
            int top = stack[--sp];
            GBVHNode node = getBVHNode(top);
            if (node.n > 0) {
                cnt->leaves++;
                int L = node.index;
                int R = node.index + node.n - 1;
                HitResult r = hitArray(ray, L, R);
                if (r.isHit && r.distance < res.distance) res = r;
                continue;
            }
            cnt->internal++;
            float d1 = INF;
            float d2 = INF;
            if (node.left > 0) {
                GBVHNode leftNode = getBVHNode(node.left);
                d1 = hitAABB(ray, leftNode.AA, leftNode.BB);
            }
            if (node.right > 0) {
                GBVHNode rightNode = getBVHNode(node.right);
                d2 = hitAABB(ray, rightNode.AA, rightNode.BB);
            }
            if (d1 > 0 && d2 > 0) {
                if (d1 < d2) {
                    stack[sp++] = node.right;
                    stack[sp++] = node.left;
                } else {
                    stack[sp++] = node.left;
                    stack[sp++] = node.right;
                }
            } else if (d1 > 0) {
                stack[sp++] = node.left;
            } else if (d2 > 0) {
                stack[sp++] = node.right;
            }
        }
        return res;
    }

    // f5:320-336
    void initSeed() {
        seed = uint32_t(uint32_t((pixx * 0.5f + 0.5f) * width) * uint32_t(1973) + uint32_t((pixy * 0.5f + 0.5f) * height) * uint32_t(9277) +
                        uint32_t(frameCounter) * uint32_t(26699)) | uint32_t(1);
    }
    static uint32_t wang_hash(uint32_t& seed) {
        seed = uint32_t(seed ^ uint32_t(61)) ^ uint32_t(seed >> uint32_t(16));
        seed *= uint32_t(9);
        seed = seed ^ (seed >> 4);
        seed *= uint32_t(0x27d4eb2d);
        seed = seed ^ (seed >> 15);
        return seed;
    }
    float rand() { return float(wang_hash(seed)) / 4294967296.0f; }

    // f5:341-383
    static int grayCode(int i) { return i ^ (i >> 1); }
    static float sobol(int d, int i) {
        uint32_t result = 0u;
        int offset = 32 * d;
        for (int j = 0; i != 0; i >>= 1, j++)
            if ((i & 1) != 0) result ^= V[offset + j];
        return float(result) * (1.0f / float(0xFFFFFFFFU));
    }
    static void sobolVec2(int i, int b, float* u, float* v) {
        *u = sobol(b * 2, grayCode(i));
        *v = sobol(b * 2 + 1, grayCode(i));
    }
    void CranleyPattersonRotation(float* px, float* py) const {
        uint32_t pseed = uint32_t(uint32_t((pixx * 0.5f + 0.5f) * width) * uint32_t(1973) + uint32_t((pixy * 0.5f + 0.5f) * height) * uint32_t(9277) +
                                  uint32_t(114514 / 1919) * uint32_t(26699)) | uint32_t(1);
        float u = float(wang_hash(pseed)) / 4294967296.0f;
        float v = float(wang_hash(pseed)) / 4294967296.0f;
        *px += u;
        if (*px > 1) *px -= 1;
        if (*px < 0) *px += 1;
        *py += v;
        if (*py > 1) *py -= 1;
        if (*py < 0) *py += 1;
    }

    // f5:387-428
    static float sqr(float x) { return x * x; }
    static float SchlickFresnel(float u) {
        float m = clamp(1 - u, 0, 1);
        float m2 = m * m;
        return m2 * m2 * m;
    }
    static float GTR1(float NdotH, float a) {
        if (a >= 1) return 1 / PI;
        float a2 = a * a;
        float t = 1 + (a2 - 1) * NdotH * NdotH;
        return (a2 - 1) / (PI * M::log_(a2) * t);
    }
    static float GTR2(float NdotH, float a) {
        float a2 = a * a;
        float t = 1 + (a2 - 1) * NdotH * NdotH;
        return a2 / (PI * t * t);
    }
    static float GTR2_aniso(float NdotH, float HdotX, float HdotY, float ax, float ay) {
        return 1 / (PI * ax * ay * sqr(sqr(HdotX / ax) + sqr(HdotY / ay) + NdotH * NdotH));
    }
    static float smithG_GGX(float NdotV, float alphaG) {
        float a = alphaG * alphaG;
        float b = NdotV * NdotV;
        return 1 / (NdotV + sqrtf(a + b - a * b));
    }
    static float smithG_GGX_aniso(float NdotV, float VdotX, float VdotY, float ax, float ay) {
        return 1 / (NdotV + sqrtf(sqr(VdotX * ax) + sqr(VdotY * ay) + sqr(NdotV)));
    }

    // f5:430-482 (aniso == false) and f4:214-278 (aniso == true)
    static vec3 BRDF_Evaluate(vec3 V, vec3 N, vec3 L, vec3 X, vec3 Y, const Material& material, bool aniso) {
        float NdotL = dot(N, L);
        float NdotV = dot(N, V);
        if (NdotL < 0 || NdotV < 0) return vec3(0);
        vec3 H = normalize(L + V);
        float NdotH = dot(N, H);
        float LdotH = dot(L, H);
        vec3 Cdlin = material.baseColor;
        float Cdlum = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;
        vec3 Ctint = (Cdlum > 0) ? (Cdlin / Cdlum) : (vec3(1));
        vec3 Cspec = material.specular * mix(vec3(1), Ctint, material.specularTint);
        vec3 Cspec0 = mix(0.08f * Cspec, Cdlin, material.metallic);
        vec3 Csheen = mix(vec3(1), Ctint, material.sheenTint);
        float Fd90 = 0.5f + 2.0f * material.roughness * LdotH * LdotH;
        float FL = SchlickFresnel(NdotL);
        float FV = SchlickFresnel(NdotV);
        float Fd = mix(1.0f, Fd90, FL) * mix(1.0f, Fd90, FV);
        float Fss90 = LdotH * LdotH * material.roughness;
        float Fss = mix(1.0f, Fss90, FL) * mix(1.0f, Fss90, FV);
        float ss = 1.25f * (Fss * (1.0f / (NdotL + NdotV) - 0.5f) + 0.5f);
        float Ds, Gs;
        float FH = SchlickFresnel(LdotH);
        vec3 Fs = mix(Cspec0, vec3(1), FH);
        if (aniso) {
            float aspect = sqrtf(1.0f - material.anisotropic * 0.9f);
            float ax = max(0.001f, sqr(material.roughness) / aspect);
            float ay = max(0.001f, sqr(material.roughness) * aspect);
            Ds = GTR2_aniso(NdotH, dot(H, X), dot(H, Y), ax, ay);
            Gs = smithG_GGX_aniso(NdotL, dot(L, X), dot(L, Y), ax, ay);
            Gs *= smithG_GGX_aniso(NdotV, dot(V, X), dot(V, Y), ax, ay);
        } else {
            float alpha = max(0.001f, sqr(material.roughness));
            Ds = GTR2(NdotH, alpha);
            Gs = smithG_GGX(NdotL, material.roughness);
            Gs *= smithG_GGX(NdotV, material.roughness);
        }
        float Dr = GTR1(NdotH, mix(0.1f, 0.001f, material.clearcoatGloss));
        float Fr = mix(0.04f, 1.0f, FH);
        float Gr = smithG_GGX(NdotL, 0.25f) * smithG_GGX(NdotV, 0.25f);
        vec3 Fsheen = FH * material.sheen * Csheen;
        vec3 diffuse = (1.0f / PI) * mix(Fd, ss, material.subsurface) * Cdlin + Fsheen;
        vec3 specular = Gs * Fs * Ds;
        vec3 clearcoat = vec3(0.25f * Gr * Fr * Dr * material.clearcoat);
        return diffuse * (1.0f - material.metallic) + specular + clearcoat;
    }

    // f5:486-501
    static void getTagent(vec3 N, vec3& tangent, vec3& bitangent) {
        vec3 helper = vec3(1, 0, 0);
        if (fabsf(N.x) > 0.999f) helper = vec3(0, 0, 1);
        bitangent = normalize(cross(N, helper));
        tangent = normalize(cross(N, bitangent));
    }
    static vec3 toNormalHemisphere(vec3 v, vec3 N) {
        vec3 helper = vec3(1, 0, 0);
        if (fabsf(N.x) > 0.999f) helper = vec3(0, 0, 1);
        vec3 tangent = normalize(cross(N, helper));
        vec3 bitangent = normalize(cross(N, tangent));
        return v.x * tangent + v.y * bitangent + v.z * N;
    }

    // f4:110-115
    vec3 SampleHemisphere() {
        float z = rand();
        float r = max(0, sqrtf(1.0f - z * z));
        float phi = 2.0f * PI * rand();
        return vec3(r * M::cos_(phi), r * M::sin_(phi), z);
    }
    // f5:524-608
    static vec3 SampleCosineHemisphere(float xi_1, float xi_2, vec3 N) {
        float r = sqrtf(xi_1);
        float theta = xi_2 * 2.0f * PI;
        float x = r * M::cos_(theta);
        float y = r * M::sin_(theta);
        float z = sqrtf(1.0f - x * x - y * y);
        return toNormalHemisphere(vec3(x, y, z), N);
    }
    static vec3 SampleGTR1(float xi_1, float xi_2, vec3 V, vec3 N, float alpha) {
        float phi_h = 2.0f * PI * xi_1;
        float sin_phi_h = M::sin_(phi_h);
        float cos_phi_h = M::cos_(phi_h);
        float cos_theta_h = sqrtf((1.0f - M::pow_(alpha * alpha, 1.0f - xi_2)) / (1.0f - alpha * alpha));
        float sin_theta_h = sqrtf(max(0.0f, 1.0f - cos_theta_h * cos_theta_h));
        vec3 H = vec3(sin_theta_h * cos_phi_h, sin_theta_h * sin_phi_h, cos_theta_h);
        H = toNormalHemisphere(H, N);
        return reflect(-V, H);
    }
    static vec3 SampleGTR2(float xi_1, float xi_2, vec3 V, vec3 N, float alpha) {
        float phi_h = 2.0f * PI * xi_1;
        float sin_phi_h = M::sin_(phi_h);
        float cos_phi_h = M::cos_(phi_h);
        float cos_theta_h = sqrtf((1.0f - xi_2) / (1.0f + (alpha * alpha - 1.0f) * xi_2));
        float sin_theta_h = sqrtf(max(0.0f, 1.0f - cos_theta_h * cos_theta_h));
        vec3 H = vec3(sin_theta_h * cos_phi_h, sin_theta_h * sin_phi_h, cos_theta_h);
        H = toNormalHemisphere(H, N);
        return reflect(-V, H);
    }
    static vec3 SampleBRDF(float xi_1, float xi_2, float xi_3, vec3 V, vec3 N, const Material& material) {
        float alpha_GTR1 = mix(0.1f, 0.001f, material.clearcoatGloss);
        float alpha_GTR2 = max(0.001f, sqr(material.roughness));
        float r_diffuse = (1.0f - material.metallic);
        float r_specular = 1.0f;
        float r_clearcoat = 0.25f * material.clearcoat;
        float r_sum = r_diffuse + r_specular + r_clearcoat;
        float p_diffuse = r_diffuse / r_sum;
        float p_specular = r_specular / r_sum;
        float rd = xi_3;
        if (rd <= p_diffuse) {
            return SampleCosineHemisphere(xi_1, xi_2, N);
        } else if (p_diffuse < rd && rd <= p_diffuse + p_specular) {
            return SampleGTR2(xi_1, xi_2, V, N, alpha_GTR2);
        } else if (p_diffuse + p_specular < rd) {
            return SampleGTR1(xi_1, xi_2, V, N, alpha_GTR1);
        }
        return vec3(0, 1, 0);
    }

    // f5:612-660
    vec3 SampleHdr(float xi_1, float xi_2) const {
        vec3 c = texture2D(hdrCache, xi_1, xi_2);
        float x = c.x, y = c.y;
        y = 1.0f - y;
        float phi = 2.0f * PI * (x - 0.5f);
        float theta = PI * (y - 0.5f);
        return vec3(M::cos_(theta) * M::cos_(phi), M::sin_(theta), M::cos_(theta) * M::sin_(phi));
    }
    static void toSphericalCoord(vec3 v, float* u, float* w) {
        float ux = M::atan2_(v.z, v.x), uy = M::asin_(v.y);
        ux /= (2.0f * PI), uy /= PI;
        ux += 0.5f, uy += 0.5f;
        uy = 1.0f - uy;
        *u = ux, *w = uy;
    }
    vec3 hdrColor(vec3 v) const {  // f5:638-644; chapters 3/4: sampleHdr with min(color, 10), f4:159-165
        float u, w;
        toSphericalCoord(normalize(v), &u, &w);
        vec3 color = texture2D(hdrMap, u, w);
        if (CHAP != 5) color = min(color, vec3(10));
        return color;
    }
    float hdrPdf(vec3 L, int hdrResolution) const {
        float u, w;
        toSphericalCoord(normalize(L), &u, &w);
        float pdf = texture2D(hdrCache, u, w).z;
        float theta = PI * (0.5f - w);
        float sin_theta = max(M::sin_(theta), 1e-10f);
        float p_convert = float(hdrResolution * hdrResolution / 2) / (2.0f * PI * PI * sin_theta);
        return pdf * p_convert;
    }
    static float BRDF_Pdf(vec3 V, vec3 N, vec3 L, const Material& material) {  // f5:662-699
        float NdotL = dot(N, L);
        float NdotV = dot(N, V);
        if (NdotL < 0 || NdotV < 0) return 0;
        vec3 H = normalize(L + V);
        float NdotH = dot(N, H);
        float alpha_Ds = max(0.001f, sqr(material.roughness));
        float Ds = GTR2(NdotH, alpha_Ds);
        float Dr = GTR1(NdotH, mix(0.1f, 0.001f, material.clearcoatGloss));
        float pdf_diffuse = NdotL / PI;
        float pdf_specular = Ds * NdotH / (4.0f * dot(L, H));
        float pdf_clearcoat = Dr * NdotH / (4.0f * dot(L, H));
        float r_diffuse = (1.0f - material.metallic);
        float r_specular = 1.0f;
        float r_clearcoat = 0.25f * material.clearcoat;
        float r_sum = r_diffuse + r_specular + r_clearcoat;
        float p_diffuse = r_diffuse / r_sum;
        float p_specular = r_specular / r_sum;
        float p_clearcoat = r_clearcoat / r_sum;
        float pdf = p_diffuse * pdf_diffuse + p_specular * pdf_specular + p_clearcoat * pdf_clearcoat;
        pdf = max(1e-10f, pdf);
        return pdf;
    }
    static float misMixWeight(float a, float b) {
        float t = a * a;
        return t / (b * b + t);
    }

    vec3 pathTracingImportanceSampling(HitResult hit, int maxBounce) {  // f5:850-923
        vec3 Lo = vec3(0);
        vec3 history = vec3(1);
        for (int bounce = 0; bounce < maxBounce; ++bounce) {
            vec3 V = -hit.viewDir;
            vec3 N = hit.normal;
            Ray hdrTestRay;
            hdrTestRay.startPoint = hit.hitPoint;
            float r1 = rand();  // GLSL evaluates call arguments left to right
            float r2 = rand();
            hdrTestRay.direction = SampleHdr(r1, r2);
            if (dot(N, hdrTestRay.direction) > 0.0f) {
                HitResult hdrHit = hitBVH(hdrTestRay);
                if (!hdrHit.isHit) {
                    vec3 L = hdrTestRay.direction;
                    vec3 color = hdrColor(L);
                    float pdf_light = hdrPdf(L, hdrResolution);
                    vec3 f_r = BRDF_Evaluate(V, N, L, vec3(0), vec3(0), hit.material, false);
                    float pdf_brdf = BRDF_Pdf(V, N, L, hit.material);
                    float mis_weight = misMixWeight(pdf_light, pdf_brdf);
                    Lo += mis_weight * history * color * f_r * dot(N, L) / pdf_light;
                }
            }
            float su, sv;
            sobolVec2(frameCounter + 1, bounce, &su, &sv);
            CranleyPattersonRotation(&su, &sv);
            float xi_1 = su;
            float xi_2 = sv;
            float xi_3 = rand();
            vec3 L = SampleBRDF(xi_1, xi_2, xi_3, V, N, hit.material);
            float NdotL = dot(N, L);
            if (NdotL <= 0.0f) break;
            Ray randomRay;
            randomRay.startPoint = hit.hitPoint;
            randomRay.direction = L;
            HitResult newhit = hitBVH(randomRay);
            vec3 f_r = BRDF_Evaluate(V, N, L, vec3(0), vec3(0), hit.material, false);
            float pdf_brdf = BRDF_Pdf(V, N, L, hit.material);
            if (pdf_brdf <= 0.0f) break;
            if (!newhit.isHit) {
                vec3 color = hdrColor(L);
                float pdf_light = hdrPdf(L, hdrResolution);
                float mis_weight = misMixWeight(pdf_brdf, pdf_light);
                Lo += mis_weight * history * color * NdotL * f_r / pdf_brdf;
                break;
            }
            vec3 Le = newhit.material.emissive;
            Lo += history * Le * NdotL * f_r / pdf_brdf;
            hit = newhit;
            history *= NdotL * f_r / pdf_brdf;
        }
        return Lo;
    }

    vec3 pathTracing4(HitResult hit, int maxBounce) {  // f4:505-541
        vec3 Lo = vec3(0);
        vec3 history = vec3(1);
        for (int bounce = 0; bounce < maxBounce; ++bounce) {
            vec3 V = -hit.viewDir;
            vec3 N = hit.normal;
            vec3 L = toNormalHemisphere(SampleHemisphere(), hit.normal);
            float cosine_i = max(0, dot(N, L));
            float pdf = 1.0f / (2.0f * PI);
            vec3 tangent, bitangent;
            getTagent(N, tangent, bitangent);
            vec3 f_r = BRDF_Evaluate(V, N, L, tangent, bitangent, hit.material, true);
            Ray randomRay;
            randomRay.startPoint = hit.hitPoint;
            randomRay.direction = L;
            HitResult newhit = hitBVH(randomRay);
            if (!newhit.isHit) {
                vec3 skyColor = hdrColor(randomRay.direction);
                Lo += history * skyColor * f_r * cosine_i / pdf;
                break;
            }
            vec3 Le = newhit.material.emissive;
            Lo += history * Le * cosine_i / pdf;
            hit = newhit;
            history *= f_r * cosine_i / pdf;
        }
        return Lo;
    }

    vec3 pathTracing3(HitResult hit, int maxBounce) {  // f3:378-416
        vec3 Lo = vec3(0);
        vec3 history = vec3(1);
        for (int bounce = 0; bounce < maxBounce; bounce++) {
            vec3 wi = toNormalHemisphere(SampleHemisphere(), hit.normal);
            Ray randomRay;
            randomRay.startPoint = hit.hitPoint;
            randomRay.direction = wi;
            HitResult newHit = hitBVH(randomRay);
            float pdf = 1.0f / (2.0f * PI);
            float cosine_i = max(0, dot(randomRay.direction, hit.normal));
            vec3 f_r = hit.material.baseColor / PI;
            if (!newHit.isHit) {
                vec3 skyColor = hdrColor(randomRay.direction);
                Lo += history * skyColor * f_r * cosine_i / pdf;
                break;
            }
            vec3 Le = newHit.material.emissive;
            Lo += history * Le * f_r * cosine_i / pdf;
            hit = newHit;
            history *= f_r * cosine_i / pdf;
        }
        return Lo;
    }

    // f5:926-988 without the blend with lastFrame
    vec3 mainColor(int maxBounce) {
        initSeed();
        Ray ray;
        ray.startPoint = eye;
        float a1 = rand();
        float a2 = rand();
        float AAx = (a1 - 0.5f) / float(width), AAy = (a2 - 0.5f) / float(height);
        float v[4] = {pixx + AAx, pixy + AAy, -1.5f, 0.0f}, dir[4];
        mulv(cameraRotate, v, dir);
        ray.direction = normalize(vec3(dir[0], dir[1], dir[2]));
        HitResult firstHit = hitBVH(ray);
        vec3 color;
        if (!firstHit.isHit) {
            color = hdrColor(ray.direction);
        } else {
            vec3 Le = firstHit.material.emissive;
            vec3 Li = (CHAP == 5) ? pathTracingImportanceSampling(firstHit, maxBounce)
                                  : ((CHAP == 4) ? pathTracing4(firstHit, maxBounce) : pathTracing3(firstHit, maxBounce));
            color = Le + Li;
        }
        return color;
    }
};

template <class M, int CHAP>
static void render_t(const float* tris, int nT, const float* nodes, int nN, const float* hdr, const float* cache, int hw, int hh, int width,
                     int height, int maxBounce, int accumulate, const float* eye, const float* cam, int frame_begin, int frame_count,
                     float* accum, uint64_t* stats) {
    Counters total;
#pragma omp parallel
    {
        Counters cnt;
#pragma omp for schedule(dynamic, 1)
        for (int py = 0; py < height; ++py) {
            for (int px = 0; px < width; ++px) {
                Shader<M, CHAP> sh;
                sh.nTriangles = nT, sh.nNodes = nN, sh.width = width, sh.height = height, sh.hdrResolution = hw;
                sh.triangles = tris, sh.nodes = nodes, sh.hdrMap = hdr, sh.hdrCache = cache, sh.hdrW = hw, sh.hdrH = hh;
                sh.eye = vec3(eye[0], eye[1], eye[2]);
                memcpy(sh.cameraRotate.m, cam, 64);
                sh.cnt = &cnt;
                // pix: the quad position interpolated at the pixel centre (vshader.vsh:7-10); row 0 = bottom
                sh.pixx = -1.0f + 2.0f * ((float)px + 0.5f) / (float)width;
                sh.pixy = -1.0f + 2.0f * ((float)py + 0.5f) / (float)height;
                float* o = accum + ((size_t)py * width + px) * 3;
                vec3 last(o[0], o[1], o[2]);
                for (int f = 0; f < frame_count; ++f) {
                    sh.frameCounter = frame_begin + f;
                    vec3 color = sh.mainColor(maxBounce);
                    cnt.paths++;
                    if (accumulate == 1)
                        last = last + color;
                    else
                        last = mix(last, color, 1.0f / (sh.frameCounter + 1));  // f5:955
                }
                o[0] = last.x, o[1] = last.y, o[2] = last.z;
            }
        }
#pragma omp critical
        {
            total.paths += cnt.paths, total.rays += cnt.rays, total.internal += cnt.internal;
            total.leaves += cnt.leaves, total.tris += cnt.tris, total.updates += cnt.updates;
        }
    }
    if (stats) stats[0] = total.paths, stats[1] = total.rays, stats[2] = total.internal, stats[3] = total.leaves, stats[4] = total.tris, stats[5] = total.updates;
}

// ---------------------------------------------------------------------------------------------------------------
// Chapter 2 closest hit on the flattened arrays: hitTriangle c2:230-256, hitTriangleArray c2:451-464, hitAABB
// c2:467-487, hitBVH c2:490-513 (recursive; r1.distance < r2.distance ? r1 : r2).
// ---------------------------------------------------------------------------------------------------------------
struct C2Hit {
    int triangle = -1;
    float distance = 114514.0f;
};
struct C2 {
    const float* tris;
    const float* nodes;
    vec3 P3(int tri, int k) const { return vec3(tris[tri * 36 + 3 * k], tris[tri * 36 + 3 * k + 1], tris[tri * 36 + 3 * k + 2]); }
    float hitTriangle(int tri, const Ray& ray) const {
        vec3 p1 = P3(tri, 0), p2 = P3(tri, 1), p3 = P3(tri, 2);
        vec3 S = ray.startPoint, d = ray.direction;
        vec3 N = normalize(cross(p2 - p1, p3 - p1));
        if (dot(N, d) > 0.0f) N = -N;
        if (fabsf(dot(d, N)) < 0.00001f) return 114514.0f;
        float t = (-dot(S, N) + dot(N, p1)) / dot(d, N);
        if (t < 0.0005f) return 114514.0f;
        vec3 P = S + d * t;
        vec3 c1 = cross(p2 - p1, P - p1);
        vec3 c2 = cross(p3 - p2, P - p2);
        vec3 c3 = cross(p1 - p3, P - p3);
        if (dot(c1, N) > 0 && dot(c2, N) > 0 && dot(c3, N) > 0) return t;
        if (dot(c1, N) < 0 && dot(c2, N) < 0 && dot(c3, N) < 0) return t;
        return 114514.0f;
    }
    C2Hit hitTriangleArray(const Ray& ray, int l, int r) const {
        C2Hit res;
        for (int i = l; i <= r; ++i) {
            float d = hitTriangle(i, ray);
            if (d < 114514.0f && d < res.distance) {
                res.distance = d;
                res.triangle = i;
            }
        }
        return res;
    }
    static float hitAABB(const Ray& r, vec3 AA, vec3 BB) {
        vec3 invdir = vec3(1.0 / r.direction.x, 1.0 / r.direction.y, 1.0 / r.direction.z);
        vec3 in = (BB - r.startPoint) * invdir;
        vec3 out = (AA - r.startPoint) * invdir;
        vec3 tmax = max(in, out);
        vec3 tmin = min(in, out);
        float t1 = min(tmax.x, min(tmax.y, tmax.z));
        float t0 = max(tmin.x, max(tmin.y, tmin.z));
        return (t1 >= t0) ? ((t0 > 0.0) ? (t0) : (t1)) : (-1);
    }
    C2Hit hitBVH(const Ray& ray, int root) const {
        if (root == 0) return C2Hit();
        const float* nd = nodes + (size_t)root * 12;
        int left = (int)nd[0], right = (int)nd[1], n = (int)nd[3], index = (int)nd[4];
        if (n > 0) return hitTriangleArray(ray, index, n + index - 1);
        float d1 = 114514.0f, d2 = 114514.0f;
        if (left) d1 = hitAABB(ray, vec3(nodes[left * 12 + 6], nodes[left * 12 + 7], nodes[left * 12 + 8]), vec3(nodes[left * 12 + 9], nodes[left * 12 + 10], nodes[left * 12 + 11]));
        if (right) d2 = hitAABB(ray, vec3(nodes[right * 12 + 6], nodes[right * 12 + 7], nodes[right * 12 + 8]), vec3(nodes[right * 12 + 9], nodes[right * 12 + 10], nodes[right * 12 + 11]));
        C2Hit r1, r2;
        if (d1 > 0) r1 = hitBVH(ray, left);
        if (d2 > 0) r2 = hitBVH(ray, right);
        return r1.distance < r2.distance ? r1 : r2;
    }
};

}  // namespace orc

using namespace orc;

extern "C" {

struct orc_mesh {
    std::vector<Triangle> tris;
};
static_assert(sizeof(Triangle) == 144, "Triangle must match Triangle_encoded");

orc_mesh* orc_mesh_new() { return new orc_mesh(); }
void orc_mesh_free(orc_mesh* m) { delete m; }
int orc_mesh_count(orc_mesh* m) { return (int)m->tris.size(); }
void orc_mesh_copy(orc_mesh* m, float* out36) {  // encode, c5:848-867
    for (size_t i = 0; i < m->tris.size(); ++i) {
        const Triangle& t = m->tris[i];
        const Material& mt = t.material;
        float* o = out36 + i * 36;
        const vec3 v[12] = {t.p1, t.p2, t.p3, t.n1, t.n2, t.n3, mt.emissive, mt.baseColor, vec3(mt.subsurface, mt.metallic, mt.specular),
                            vec3(mt.specularTint, mt.roughness, mt.anisotropic), vec3(mt.sheen, mt.sheenTint, mt.clearcoat),
                            vec3(mt.clearcoatGloss, mt.IOR, mt.transmission)};
        for (int k = 0; k < 12; ++k) o[3 * k] = v[k].x, o[3 * k + 1] = v[k].y, o[3 * k + 2] = v[k].z;
    }
}
void orc_transform_matrix(const float* rot, const float* tr, const float* sc, float* out16) {
    mat4 m = getTransformMatrix(vec3(rot[0], rot[1], rot[2]), vec3(tr[0], tr[1], tr[2]), vec3(sc[0], sc[1], sc[2]));
    memcpy(out16, m.m, 64);
}
// mat18: emissive, baseColor, subsurface, metallic, specular, specularTint, roughness, anisotropic, sheen, sheenTint, clearcoat, clearcoatGloss, IOR, transmission
int orc_mesh_read_obj(orc_mesh* m, const char* path, const float* mat18, const float* trans16, int smooth) {
    Material mt;
    mt.emissive = vec3(mat18[0], mat18[1], mat18[2]), mt.baseColor = vec3(mat18[3], mat18[4], mat18[5]);
    mt.subsurface = mat18[6], mt.metallic = mat18[7], mt.specular = mat18[8], mt.specularTint = mat18[9], mt.roughness = mat18[10];
    mt.anisotropic = mat18[11], mt.sheen = mat18[12], mt.sheenTint = mat18[13], mt.clearcoat = mat18[14], mt.clearcoatGloss = mat18[15];
    mt.IOR = mat18[16], mt.transmission = mat18[17];
    mat4 t;
    memcpy(t.m, trans16, 64);
    return readObj(path, m->tris, mt, t, smooth != 0);
}
int orc_mesh_read_obj_chap2(orc_mesh* m, const char* path, float scl, float yoff) { return readObjChap2(path, m->tris, scl, yoff); }
// returns the node count (dummy node 0 included, c5:835-843); nodes_out holds cap BVHNode_encoded records
int orc_bvh_build(orc_mesh* m, int leaf_n, int use_sah, float* nodes_out, int cap) {
    BVHNode testNode;
    memset((void*)&testNode, 0, sizeof(testNode));
    std::vector<BVHNode> nodes{testNode};
    if (use_sah)
        buildBVHwithSAH(m->tris, nodes, 0, (int)m->tris.size() - 1, leaf_n);
    else
        buildBVH(m->tris, nodes, 0, (int)m->tris.size() - 1, leaf_n);
    if ((int)nodes.size() > cap) return -(int)nodes.size();
    for (size_t i = 0; i < nodes.size(); ++i) {  // c5:869-875
        float* o = nodes_out + i * 12;
        o[0] = nodes[i].left, o[1] = nodes[i].right, o[2] = 0;
        o[3] = nodes[i].n, o[4] = nodes[i].index, o[5] = 0;
        o[6] = nodes[i].AA.x, o[7] = nodes[i].AA.y, o[8] = nodes[i].AA.z;
        o[9] = nodes[i].BB.x, o[10] = nodes[i].BB.y, o[11] = nodes[i].BB.z;
    }
    return (int)nodes.size();
}
int orc_hdr_load(const char* fileName, int* w_out, int* h_out, float** cols_out) {  // hdr:29-97
    FILE* file = fopen(fileName, "rb");
    if (!file) return -1;
    char str[200];
    if (fread(str, 10, 1, file) != 1 || memcmp(str, "#?RADIANCE", 10)) {
        fclose(file);
        return -2;
    }
    fseek(file, 1, SEEK_CUR);
    int i = 0;
    int c = 0, oldc;
    while (true) {
        oldc = c;
        c = fgetc(file);
        if (c == EOF) return fclose(file), -2;
        if (c == 0xa && oldc == 0xa) break;
    }
    char reso[200];
    i = 0;
    while (true) {
        c = fgetc(file);
        if (c == EOF || i > 190) return fclose(file), -2;
        reso[i++] = c;
        if (c == 0xa) break;
    }
    reso[i] = 0;
    int w, h;
    if (sscanf(reso, "-Y %d +X %d", &h, &w) != 2) return fclose(file), -2;
    float* cols = (float*)calloc((size_t)w * h * 3, sizeof(float));
    *cols_out = cols;
    RGBE* scanline = new RGBE[w];
    for (int y = h - 1; y >= 0; y--) {
        if (decrunch(scanline, w, file) == false) break;
        for (int x = 0; x < w; ++x) {  // workOnRGBE, hdr:106-116
            int expo = scanline[x][3] - 128;
            cols[0] = convertComponent(expo, scanline[x][0]);
            cols[1] = convertComponent(expo, scanline[x][1]);
            cols[2] = convertComponent(expo, scanline[x][2]);
            cols += 3;
        }
    }
    delete[] scanline;
    fclose(file);
    *w_out = w, *h_out = h;
    return 0;
}
void orc_free(void* p) { free(p); }
void orc_hdr_cache(const float* hdr, int w, int h, float* out) {
    float* c = calculateHdrCache(hdr, w, h);
    memcpy(out, c, (size_t)w * h * 3 * sizeof(float));
    delete[] c;
}
void orc_camera(float rotatAngle, float upAngle, float r, float* eye_out, float* cam16) {  // c5:716-720
    vec3 eye = vec3(-sinf(radians(rotatAngle)) * cosf(radians(upAngle)), sinf(radians(upAngle)), cosf(radians(rotatAngle)) * cosf(radians(upAngle)));
    eye.x *= r, eye.y *= r, eye.z *= r;
    mat4 cameraRotate = lookAt(eye, vec3(0, 0, 0), vec3(0, 1, 0));
    cameraRotate = inverse(cameraRotate);
    eye_out[0] = eye.x, eye_out[1] = eye.y, eye_out[2] = eye.z;
    memcpy(cam16, cameraRotate.m, 64);
}

// mode 3/4/5; accumulate 0 = running mean (f5:955), 1 = sum; use_libm: transcendental policy; stats6 may be NULL
int orc_render(const float* tris, int nT, const float* nodes, int nN, const float* hdr, const float* cache, int hw, int hh, int width, int height,
               int mode, int maxBounce, int accumulate, const float* eye, const float* cam, int frame_begin, int frame_count, float* accum,
               int use_libm, uint64_t* stats6) {
    if (maxBounce <= 0) maxBounce = (mode == 5) ? 4 : 6;
#define ORC_CALL(MATH, CH) render_t<MATH, CH>(tris, nT, nodes, nN, hdr, cache, hw, hh, width, height, maxBounce, accumulate, eye, cam, frame_begin, frame_count, accum, stats6)
    if (use_libm) {
        if (mode == 5) ORC_CALL(LibmMath, 5);
        else if (mode == 4) ORC_CALL(LibmMath, 4);
        else if (mode == 3) ORC_CALL(LibmMath, 3);
        else return -1;
    } else {
        if (mode == 5) ORC_CALL(DetMath, 5);
        else if (mode == 4) ORC_CALL(DetMath, 4);
        else if (mode == 3) ORC_CALL(DetMath, 3);
        else return -1;
    }
#undef ORC_CALL
    return 0;
}

// chapter-2 closest hit for the un-jittered camera rays
void orc_primary_hits(const float* tris, int nT, const float* nodes, int nN, int width, int height, const float* eye, const float* cam,
                      int32_t* tri_out, float* dist_out) {
    C2 c2{tris, nodes};
    mat4 cr;
    memcpy(cr.m, cam, 64);
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < height; ++py)
        for (int px = 0; px < width; ++px) {
            float v[4] = {-1.0f + 2.0f * ((float)px + 0.5f) / (float)width, -1.0f + 2.0f * ((float)py + 0.5f) / (float)height, -1.5f, 0.0f}, dir[4];
            mulv(cr, v, dir);
            Ray ray;
            ray.startPoint = vec3(eye[0], eye[1], eye[2]);
            ray.direction = normalize(vec3(dir[0], dir[1], dir[2]));
            C2Hit h = (nN >= 2) ? c2.hitBVH(ray, 1) : C2Hit();
            tri_out[(size_t)py * width + px] = h.triangle;
            dist_out[(size_t)py * width + px] = h.distance;
        }
}
// one ray (the c2:605-609 known-answer test)
void orc_hit_one(const float* tris, const float* nodes, const float* origin, const float* dir_unnormalised, int32_t* tri_out, float* dist_out) {
    C2 c2{tris, nodes};
    Ray ray;
    ray.startPoint = vec3(origin[0], origin[1], origin[2]);
    ray.direction = normalize(vec3(dir_unnormalised[0], dir_unnormalised[1], dir_unnormalised[2]));
    C2Hit h = c2.hitBVH(ray, 1);
    *tri_out = h.triangle, *dist_out = h.distance;
}

// pass3.fsh:14-25 + unorm8 conversion
void orc_tonemap_rgb8(const float* rgb, int width, int height, unsigned char* out, int flip_y) {
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const float* c = rgb + ((size_t)y * width + x) * 3;
            float luminance = 0.3f * c[0] + 0.6f * c[1] + 0.1f * c[2];
            unsigned char* o = out + ((size_t)(flip_y ? height - 1 - y : y) * width + x) * 3;
            for (int k = 0; k < 3; ++k) {
                float v = c[k] * 1.0f / (1.0f + luminance / 1.5f);
                v = powf(v, 1.0f / 2.2f);
                v = (v >= 0.0f) ? ((v <= 1.0f) ? v : 1.0f) : 0.0f;
                o[k] = (unsigned char)(int)floorf(v * 255.0f + 0.5f);
            }
        }
}

// deterministic-math probes for tests
float orc_det_sin(float x) { return det_sin(x); }
float orc_det_cos(float x) { return det_cos(x); }
float orc_det_atan2(float y, float x) { return det_atan2(y, x); }
float orc_det_asin(float x) { return det_asin(x); }
float orc_det_log(float x) { return det_log(x); }
float orc_det_pow(float x, float y) { return det_pow(x, y); }
float orc_sobol(int d, int i) { return Shader<DetMath, 5>::sobol(d, i); }
uint32_t orc_wang_hash(uint32_t* seed) { return Shader<DetMath, 5>::wang_hash(*seed); }

}  // extern "C"
