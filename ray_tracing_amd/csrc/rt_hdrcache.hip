// calculateHdrCache (path_tracing_5chap.cpp:595-698) on the GPU, bit-identical to the host form (rt_hdr_cache, host_scene.cpp) and
// so to the reference's arithmetic: SURVEY 8(f)4, the feasible half (the SAH builder stays on the host: the permutation std::sort
// leaves among equal centroid coordinates is sequential by nature, DESIGN.md section 5 "Start-up").
// What is sequential in the reference stays sequential here, in the reference's order:
//   * lumSum — ONE fp32 running sum over all texels in row-major order (c5:602-611): one wave, 64 texels per coalesced load, added
//     lane by lane (v_readlane) — 524 288 dependent adds for the bundled 1024 x 512 map, ~2 ms;
//   * the marginal pdf of a column and its conditional CDF (c5:622-651): one lane per column, top to bottom;
//   * the marginal CDF over the columns (c5:628-631): one lane.
// Everything else is per texel: the luminance (evaluated in double as written, c5:607), the division by lumSum (c5:614-618) and
// the two std::lower_bound searches (c5:665-682, libstdc++'s bisection restated step for step, so that maps whose CDFs hold NaN —
// an all-black column — give the host's answer too).  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "rt_host.h"

namespace rta {

__global__ void __launch_bounds__(256) hc_lum_kernel(const float* __restrict__ hdr, float* __restrict__ pdf, size_t n) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const float R = hdr[3 * k], G = hdr[3 * k + 1], B = hdr[3 * k + 2];
    pdf[k] = (float)(0.2 * R + 0.7 * G + 0.1 * B);  // double arithmetic, rounded once (c5:607)
}

__global__ void __launch_bounds__(64) hc_lumsum_kernel(const float* __restrict__ pdf, size_t n, float* __restrict__ out) {
    const int lane = threadIdx.x;
    float sum = 0.0f;
    for (size_t base = 0; base < n; base += 64) {
        const int cnt = (n - base < 64) ? (int)(n - base) : 64;  // wave-uniform
        const float v = (lane < cnt) ? pdf[base + lane] : 0.0f;
#pragma unroll
        for (int l = 0; l < 64; ++l)
            if (l < cnt) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));  // row-major order, one add at a time
    }
    if (lane == 0) *out = sum;
}

__global__ void __launch_bounds__(256) hc_normalize_kernel(float* __restrict__ pdf, size_t n, const float* __restrict__ lum_sum) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    pdf[k] = pdf[k] / *lum_sum;  // c5:616
}

// one lane per column j: pdf_x_margin[j] (c5:622-626) and the conditional CDF of the rows given that column (c5:634-651),
// stored transposed, cdf_y[j * H + i], as the reference's cdf_y_condiciton[j][i]
__global__ void __launch_bounds__(64) hc_column_kernel(const float* __restrict__ pdf, int W, int H, float* __restrict__ margin, float* __restrict__ cdf_y) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= W) return;
    float m = 0.0f;
    for (int i = 0; i < H; ++i) m += pdf[(size_t)i * W + j];
    margin[j] = m;
    float acc = 0.0f;
    for (int i = 0; i < H; ++i) {
        const float p = pdf[(size_t)i * W + j] / m;
        acc = (i == 0) ? p : (p + acc);  // c5:643: cdf[i] += cdf[i-1]
        cdf_y[(size_t)j * H + i] = acc;
    }
}

__global__ void hc_cdfx_kernel(const float* __restrict__ margin, int W, float* __restrict__ cdf_x) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float acc = margin[0];
    cdf_x[0] = acc;
    for (int i = 1; i < W; ++i) {
        acc = margin[i] + acc;  // c5:630: cdf[i] += cdf[i-1]
        cdf_x[i] = acc;
    }
}

// std::lower_bound (libstdc++ bits/stl_algobase.h, __lower_bound): the bisection itself, so that sequences that are not sorted
// under < (NaN entries) are searched exactly as the host searches them
__device__ __forceinline__ int hc_lower_bound(const float* __restrict__ a, int n, float val) {
    int first = 0, len = n;
    while (len > 0) {
        const int half = len >> 1;
        const int mid = first + half;
        if (a[mid] < val) {
            first = mid + 1;
            len = len - half - 1;
        } else {
            len = half;
        }
    }
    return first;
}

__global__ void __launch_bounds__(256) hc_texel_kernel(const float* __restrict__ pdf, const float* __restrict__ cdf_x, const float* __restrict__ cdf_y,
                                                      int W, int H, float* __restrict__ cache) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (size_t)W * H) return;
    const int i = (int)(k / (size_t)W), j = (int)(k % (size_t)W);
    const float xi_1 = float(i) / H, xi_2 = float(j) / W;  // c5:668-669
    const int x = hc_lower_bound(cdf_x, W, xi_1);
    const int xc = (x < W) ? x : W - 1;  // (the reference indexes one past the end when xi_1 exceeds the last entry: clamped, as on the host)
    const int y = hc_lower_bound(cdf_y + (size_t)xc * H, H, xi_2);
    cache[3 * k] = float(x) / W;
    cache[3 * k + 1] = float(y) / H;
    cache[3 * k + 2] = pdf[k];
}

}  // namespace rta

using namespace rta;

extern "C" int rt_hdr_cache_gpu(const float* rgb, int width, int height, float* cache_out) {
    if (!rgb || !cache_out || width <= 0 || height <= 0) return set_error(RT_ERR_INVALID_ARG, "rt_hdr_cache_gpu: bad arguments");
    if ((long long)width * height > (1ll << 28)) return set_error(RT_ERR_INVALID_ARG, "rt_hdr_cache_gpu: bad HDR size %dx%d", width, height);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_error(RT_ERR_NO_DEVICE, "rt_hdr_cache_gpu: no HIP device available");
    const size_t n = (size_t)width * height;
    float *d_hdr = nullptr, *d_pdf = nullptr, *d_cdfy = nullptr, *d_small = nullptr, *d_cache = nullptr;
    hipError_t e = hipMalloc((void**)&d_hdr, n * 12);
    if (e == hipSuccess) e = hipMalloc((void**)&d_pdf, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_cdfy, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_small, ((size_t)2 * width + 16) * 4);  // lumSum | pdf_x_margin | cdf_x_margin
    if (e == hipSuccess) e = hipMalloc((void**)&d_cache, n * 12);
    if (e == hipSuccess) e = hipMemcpy(d_hdr, rgb, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        float* d_sum = d_small;
        float* d_margin = d_small + 16;
        float* d_cdfx = d_margin + width;
        const unsigned nb = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(hc_lum_kernel, dim3(nb), dim3(256), 0, nullptr, d_hdr, d_pdf, n);
        hipLaunchKernelGGL(hc_lumsum_kernel, dim3(1), dim3(64), 0, nullptr, d_pdf, n, d_sum);
        hipLaunchKernelGGL(hc_normalize_kernel, dim3(nb), dim3(256), 0, nullptr, d_pdf, n, d_sum);
        hipLaunchKernelGGL(hc_column_kernel, dim3((unsigned)((width + 63) / 64)), dim3(64), 0, nullptr, d_pdf, width, height, d_margin, d_cdfy);
        hipLaunchKernelGGL(hc_cdfx_kernel, dim3(1), dim3(1), 0, nullptr, d_margin, width, d_cdfx);
        hipLaunchKernelGGL(hc_texel_kernel, dim3(nb), dim3(256), 0, nullptr, d_pdf, d_cdfx, d_cdfy, width, height, d_cache);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(cache_out, d_cache, n * 12, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_hdr), (void)hipFree(d_pdf), (void)hipFree(d_cdfy), (void)hipFree(d_small), (void)hipFree(d_cache);
    if (e != hipSuccess) return set_error(RT_ERR_HIP, "rt_hdr_cache_gpu: %s", hipGetErrorString(e));
    return RT_OK;
}
