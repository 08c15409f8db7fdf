// fp32 vector math with the exact operation order of the reference's math library (GLM, README.md:70)
// so that host results are bit-identical to the reference C++ and device results are bit-identical
// to the host.  Everything here must be compiled with -ffp-contract=off.
//
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   cross(a,b)    = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
//   normalize(v)  = v * (1 / sqrt(dot(v,v)))
//   min(a,b)      = (b < a) ? b : a          max(a,b) = (a < b) ? b : a      (NaN behaviour matters: f5:225-238)
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rta {

struct vec3 {
    float x, y, z;
};

RT_HD vec3 v3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD vec3 v3(float s) { return v3(s, s, s); }

RT_HD float fmin_(float a, float b) { return (b < a) ? b : a; }
RT_HD float fmax_(float a, float b) { return (a < b) ? b : a; }
RT_HD float fabs_(float a) { return fabsf(a); }
RT_HD float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
RT_HD float mix_(float x, float y, float a) { return x * (1.0f - a) + y * a; }

RT_HD vec3 operator+(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD vec3 operator-(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD vec3 operator*(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD vec3 operator/(vec3 a, vec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
RT_HD vec3 operator*(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_HD vec3 operator*(float s, vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
RT_HD vec3 operator/(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RT_HD vec3 operator-(vec3 a) { return v3(-a.x, -a.y, -a.z); }
RT_HD vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
RT_HD vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }

RT_HD float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD vec3 cross(vec3 a, vec3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
RT_HD vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
RT_HD vec3 min3(vec3 a, vec3 b) { return v3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
RT_HD vec3 max3(vec3 a, vec3 b) { return v3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }
RT_HD vec3 mix3(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }
// GLSL reflect(I,N) = I - 2*dot(N,I)*N
RT_HD vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }

}  // namespace rta
