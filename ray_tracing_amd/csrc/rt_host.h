// Internal declarations shared by the host scene layer and the device layer.
#pragma once
#include <stdarg.h>
#include <stdio.h>

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

// host_fastbvh.cpp: the device-private entry BVH (binned SAH) over the same triangles; never visible at the ABI
struct FastBvhNode {
    float AA[3], BB[3];
    int left, right;   // internal node: children (indices into the node array)
    int first, count;  // leaf (count > 0): triangles order[first .. first + count)
};
int build_fast_bvh(const float* tri_box, const char* use, int n_tri, std::vector<FastBvhNode>& nodes, std::vector<int>& order);

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
