// Micro-benchmark: how many cycles does one wave64 fp32 VALU instruction hold a SIMD for on gfx950, as a function of
// waves per SIMD?  (Decides whether the path-loop kernel is VALU-issue-bound.)  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(64) kpk(float* out, long long* cyc, int iters, float a, float b) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    const v2 m = {a, a};
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m));
    }
    const v2 r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * 64 + threadIdx.x] = r.x + r.y;
}

template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, long long* cyc, int iters, float a, float b) {
    const long long c0 = clock64();
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // fma
            x0 = __builtin_fmaf(x0, a, b), x1 = __builtin_fmaf(x1, a, b), x2 = __builtin_fmaf(x2, a, b), x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b), x5 = __builtin_fmaf(x5, a, b), x6 = __builtin_fmaf(x6, a, b), x7 = __builtin_fmaf(x7, a, b);
        } else if (KIND == 1) {  // mul (no fma contraction)
            x0 = x0 * a, x1 = x1 * a, x2 = x2 * a, x3 = x3 * a, x4 = x4 * a, x5 = x5 * a, x6 = x6 * a, x7 = x7 * a;
        } else {  // cndmask-ish: min
            x0 = __builtin_fminf(x0, a), x1 = __builtin_fmaxf(x1, a), x2 = __builtin_fminf(x2, b), x3 = __builtin_fmaxf(x3, b);
            x4 = __builtin_fminf(x4, a), x5 = __builtin_fmaxf(x5, a), x6 = __builtin_fminf(x6, b), x7 = __builtin_fmaxf(x7, b);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - c0;
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND>
static void run(const char* name, float* d, long long* cyc, int wps) {
    const int iters = 200000, blocks = 256 * 4 * wps;
    hipEvent_t a, b;
    (void)hipEventCreate(&a), (void)hipEventCreate(&b);
    if (KIND == 3) kpk<<<blocks, 64>>>(d, cyc, 100, 1.0001f, 0.5f); else k<KIND><<<blocks, 64>>>(d, cyc, 100, 1.0001f, 0.5f);
    (void)hipEventRecord(a);
    if (KIND == 3) kpk<<<blocks, 64>>>(d, cyc, iters, 1.0001f, 0.5f); else k<KIND><<<blocks, 64>>>(d, cyc, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double inst_per_simd = (double)iters * 8 * wps;
    long long h[64];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("  [in-kernel clock64: %.2f ticks per wave-instruction of one wave]\n", (double)h[0] / ((double)iters * 8));
    printf("%s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}

int main() {
    float* d;
    long long* cyc;
    (void)hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    (void)hipMalloc(&cyc, 256 * 4 * 8 * sizeof(long long));
    for (int w : {1, 2, 4, 8}) run<0>("fma", d, cyc, w);
    for (int w : {1, 2, 4, 8}) run<1>("mul", d, cyc, w);
    for (int w : {1, 2, 4, 8}) run<2>("minmax", d, cyc, w);
    for (int w : {1, 2, 4, 5, 8}) run<3>("pk_mul (2 fp32 per lane)", d, cyc, w);
    (void)hipFree(d);
    return 0;
}
