// Row L2-normalise + temperature scale, forward and backward (f32, HBM-bound).
// One wave per row: lanes stride the row with coalesced loads, the sum of squares is reduced
// with wave shuffles, no LDS.  Also holds the library's error string and version.
#include "pclip_common.h"
#include <string.h>

thread_local char g_pclip_err[512] = "";

extern "C" int pclip_version(void) { return 1; }
extern "C" const char* pclip_last_error(void) { return g_pclip_err; }

extern "C" int pclip_check_device(int dev) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return pclip_fail(PCLIP_EDEVICE, "hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return pclip_fail(PCLIP_EDEVICE, "device %d is %s; this library holds gfx950 code only", dev, prop.gcnArchName);
    return PCLIP_OK;
}

// x = e / max(|e|, 1e-12) * exp(t/2)           (modules/models.py:50-51)
__global__ __launch_bounds__(256) void l2norm_scale_fwd_kernel(const float* __restrict__ e, const float* __restrict__ t,
                                                               float* __restrict__ x, int B, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* er = e + (size_t)row * d;
    float ss = 0.f;
    for (int k = lane; k < d; k += 64) {
        float v = er[k];
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float denom = fmaxf(sqrtf(ss), 1e-12f);
    const float s = expf(t[0] * 0.5f);
    float* xr = x + (size_t)row * d;
    for (int k = lane; k < d; k += 64) xr[k] = er[k] / denom * s;
}

// Both towers in one launch (the fused one-GPU step): rows [0,B) are e0 -> x0, rows [B,2B) are
// e1 -> x1; block 0 also clears the scalar `zero_me` (the step's dt accumulator).
__global__ __launch_bounds__(256) void l2norm_scale_fwd_pair_kernel(const float* __restrict__ e0, const float* __restrict__ e1,
                                                                    const float* __restrict__ t, float* __restrict__ x0,
                                                                    float* __restrict__ x1, int B, int d, float* __restrict__ zero_me) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && zero_me) zero_me[0] = 0.f;
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 2 * B) return;
    const bool second = row >= B;
    if (second) row -= B;
    const float* er = (second ? e1 : e0) + (size_t)row * d;
    float ss = 0.f;
    for (int k = lane; k < d; k += 64) {
        float v = er[k];
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float denom = fmaxf(sqrtf(ss), 1e-12f);
    const float s = expf(t[0] * 0.5f);
    float* xr = (second ? x1 : x0) + (size_t)row * d;
    for (int k = lane; k < d; k += 64) xr[k] = er[k] / denom * s;
}

// de = (dx - n (n.dx)) * s/|e|,  dt += 0.5 * sum(dx * x)    with n = e/|e|, x = s n.
// dx is the sum of nslab partial slabs, times the optional scalar g.
__device__ __forceinline__ void l2norm_scale_bwd_rows(const float* __restrict__ e, const float* __restrict__ t,
                                                      const float* __restrict__ dx, int nslab, size_t slab_stride,
                                                      const float* __restrict__ g, float* __restrict__ de,
                                                      float* __restrict__ dt, int B, int d, int row) {
    __shared__ float dt_part[4];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    float dts = 0.f;
    if (row >= 0 && row < B) {
        const float gs = g ? g[0] : 1.f;
        const float* er = e + (size_t)row * d;
        const float* dr = dx + (size_t)row * d;
        float ss = 0.f, dot = 0.f;
        for (int k = lane; k < d; k += 64) {
            float v = er[k];
            float a = 0.f;
            for (int s = 0; s < nslab; ++s) a += dr[(size_t)s * slab_stride + k];
            ss += v * v;
            dot += v * a;
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot) * gs;  // e . dx
        const float nrm = sqrtf(ss);
        const float s = expf(t[0] * 0.5f);
        float* der = de + (size_t)row * d;
        if (nrm >= 1e-12f) {
            const float inv = 1.f / nrm;
            const float ndx = dot * inv;  // n . dx
            for (int k = lane; k < d; k += 64) {
                float a = 0.f;
                for (int sl = 0; sl < nslab; ++sl) a += dr[(size_t)sl * slab_stride + k];
                a *= gs;
                der[k] = (a - er[k] * inv * ndx) * (s * inv);
            }
            dts = 0.5f * s * ndx;
        } else {  // norm clamped: x = e * (s / 1e-12), a plain scale
            const float sc = s / 1e-12f;
            for (int k = lane; k < d; k += 64) {
                float a = 0.f;
                for (int sl = 0; sl < nslab; ++sl) a += dr[(size_t)sl * slab_stride + k];
                der[k] = a * gs * sc;
            }
            dts = 0.5f * sc * dot;
        }
    }
    if (lane == 0) dt_part[w] = dts;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = dt_part[0] + dt_part[1] + dt_part[2] + dt_part[3];
        if (v != 0.f) atomicAdd(dt, v);
    }
}

__global__ __launch_bounds__(256) void l2norm_scale_bwd_kernel(const float* __restrict__ e, const float* __restrict__ t,
                                                               const float* __restrict__ dx, int nslab, size_t slab_stride,
                                                               const float* __restrict__ g, float* __restrict__ de,
                                                               float* __restrict__ dt, int B, int d) {
    l2norm_scale_bwd_rows(e, t, dx, nslab, slab_stride, g, de, dt, B, d, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// Both towers in one launch: blocks [0, nb) take e0/dx0/de0, blocks [nb, 2 nb) take e1/dx1/de1
// (nb = ceil(B / 4), so a block never straddles the two sides).
__global__ __launch_bounds__(256) void l2norm_scale_bwd_pair_kernel(const float* __restrict__ e0, const float* __restrict__ e1,
                                                                    const float* __restrict__ t, const float* __restrict__ dx0,
                                                                    const float* __restrict__ dx1, int nslab, size_t slab_stride,
                                                                    float* __restrict__ de0, float* __restrict__ de1,
                                                                    float* __restrict__ dt, int B, int d) {
    const int nb = (B + 3) / 4;
    const bool second = (int)blockIdx.x >= nb;
    const int row = ((int)blockIdx.x - (second ? nb : 0)) * 4 + (threadIdx.x >> 6);
    l2norm_scale_bwd_rows(second ? e1 : e0, t, second ? dx1 : dx0, nslab, slab_stride, nullptr, second ? de1 : de0, dt, B, d, row);
}

extern "C" int pclip_l2norm_scale_fwd(const float* e, const float* t, float* x, int B, int d, void* stream) {
    PCLIP_REQUIRE(e && t && x, "l2norm_scale_fwd: null pointer");
    PCLIP_REQUIRE(B >= 1 && d >= 1 && d <= 4096, "l2norm_scale_fwd: B=%d d=%d out of range", B, d);
    hipLaunchKernelGGL(l2norm_scale_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, e, t, x, B, d);
    return pclip_check_launch("l2norm_scale_fwd");
}

extern "C" int pclip_l2norm_scale_bwd(const float* e, const float* t, const float* dx, int nslab, size_t slab_stride,
                                      const float* g, float* de, float* dt, int B, int d, void* stream) {
    PCLIP_REQUIRE(e && t && dx && de && dt, "l2norm_scale_bwd: null pointer");
    PCLIP_REQUIRE(B >= 1 && d >= 1 && d <= 4096, "l2norm_scale_bwd: B=%d d=%d out of range", B, d);
    PCLIP_REQUIRE(nslab >= 1 && nslab <= 64, "l2norm_scale_bwd: nslab=%d out of range", nslab);
    hipLaunchKernelGGL(l2norm_scale_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, e, t, dx, nslab,
                       slab_stride, g, de, dt, B, d);
    return pclip_check_launch("l2norm_scale_bwd");
}


// internal: the two pair launches of the fused one-GPU CLIP step
int l2norm_scale_fwd_pair_launch(const float* e0, const float* e1, const float* t, float* x0, float* x1, int B, int d, float* zero_me,
                                 void* stream) {
    hipLaunchKernelGGL(l2norm_scale_fwd_pair_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, (hipStream_t)stream, e0, e1, t, x0, x1, B, d, zero_me);
    return pclip_check_launch("l2norm_scale_fwd_pair");
}
int l2norm_scale_bwd_pair_launch(const float* e0, const float* e1, const float* t, const float* dx0, const float* dx1, int nslab,
                                 size_t slab_stride, float* de0, float* de1, float* dt, int B, int d, void* stream) {
    hipLaunchKernelGGL(l2norm_scale_bwd_pair_kernel, dim3(2 * ((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, e0, e1, t, dx0, dx1, nslab,
                       slab_stride, de0, de1, dt, B, d);
    return pclip_check_launch("l2norm_scale_bwd_pair");
}
