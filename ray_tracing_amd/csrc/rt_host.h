// Internal declarations shared by the host scene layer and the device layer.
#pragma once
#include <stdarg.h>
#include <stdio.h>

#include <memory>
#include <utility>
#include <vector>

#include "../../include/ray_tracing_amd.h"
#include "rt_vec.h"

namespace rta {

// 36 floats, identical in memory to the reference's Triangle_encoded (c5:54-64).
struct TriRec {
    vec3 p1, p2, p3;
    vec3 n1, n2, n3;
    rt_material m;
};
static_assert(sizeof(TriRec) == 144, "Triangle_encoded is 144 bytes");

// 12 floats, identical in memory to the reference's BVHNode_encoded (c5:65-70).
struct NodeRec {
    float left, right, pad0;
    float n, index, pad1;
    vec3 AA, BB;
};
static_assert(sizeof(NodeRec) == 48, "BVHNode_encoded is 48 bytes");

int set_error(int code, const char* fmt, ...);
void add_into(float* dst, const float* src, size_t n);  // dst[i] += src[i], all host cores (host_scene.cpp)

// host_fastbvh.cpp: the device-private entry BVH (binned SAH) over the same triangles; never visible at the ABI
struct FastBvhNode {
    float AA[3], BB[3];
    int left, right;   // internal node: children (indices into the node array)
    int first, count;  // leaf (count > 0): triangles order[first .. first + count); internal node (count == 0): `first` = internal nodes in its subtree
};

// host_prep.cpp: what rt_scene_create derives from the caller's arrays before the uploads
constexpr int kMaxStack = 256;  // int stack[256], f5:265
constexpr int kDmatVec = 6;     // float4s per triangle in `dmat`: MatPre (4) + the material fields chapter 5's BRDF reads per evaluation (2)
// a leaf as a traversal entry: sign bit | (n - 1) << 24 | first triangle
constexpr int kNoChild = 0x7fffffff;  // entry of a missing child in a 4-wide entry-BVH record
inline int pack_leaf(int index, int n) { return (int)(0x80000000u | ((unsigned)(n - 1) << 24) | (unsigned)index); }
template <class T>
struct RawBuf {  // an array that is NOT zero-filled on allocation (hundreds of MB at 10 M triangles, each written once by parallel loops)
    std::unique_ptr<T[]> p;
    size_t n = 0;
    void alloc(size_t k) { p.reset(k ? new T[k] : nullptr), n = k; }
    void swap(RawBuf& o) { p.swap(o.p), std::swap(n, o.n); }
    T* data() { return p.get(); }
    const T* data() const { return p.get(); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};
int build_fast_bvh(const float* tri_box, const char* use, int n_tri, RawBuf<FastBvhNode>& nodes, int* n_nodes, std::vector<int>& order,
                   int* depth, double* sum_internal_area);
struct ScenePrep {
    bool has_root = false;
    int need_stack = 0, root_entry = 0;
    RawBuf<float> rec, ttri, dmat;           // node records (16 floats), triangle records (12), material tables (kDmatVec * 4)
    float cull_eps = 0.0f, cull_geo_max = 0.0f;
    RawBuf<float> rec2, ttri2;               // entry BVH: 4-wide records (32 floats), triangles in its leaf order
    std::vector<int> map2, parentx;          // entry-BVH triangle -> caller's index; reference record -> (parent record << 1 | side)
    std::vector<unsigned> leafx;             // per triangle: (record << 1 | side) of its reference leaf, explicit climb levels in bits 24..31
    float climb_eps = 0.0f;
    int root_entry2 = 0, stack2 = 0;         // stack2: stack entries a walk of the entry BVH can hold
    double cost_ref = 0.0, cost_entry = 0.0;
    bool has_entry = false;
    RawBuf<float> hdr4, cdir4;               // HDR texels (rgb, cache pdf); SampleHdr directions per cache texel
};
int prepare_scene(const float* tris, int nT, const float* nodes, int nN, const float* hdr_map, const float* hdr_cache, int hdrW,
                  int hdrH, int entry_mode, ScenePrep* out);

struct mat4 {
    float c[4][4];  // c[col][row], column-major like GLM
};
mat4 mat4_identity();
mat4 mat4_mul(const mat4& a, const mat4& b);
void mat4_mul_vec4(const mat4& m, const float v[4], float out[4]);

}  // namespace rta

struct rt_mesh {
    std::vector<rta::TriRec> tris;
};
