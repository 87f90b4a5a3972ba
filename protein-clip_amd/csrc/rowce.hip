// Row-wise InfoNCE with the diagonal as target (modules/training_utils_filip.py:15-23),
// forward and backward.  One wave per logit row: lanes stride the row with coalesced loads,
// keep a running (max, sum), and merge with wave shuffles.
#include "pclip_common.h"

__global__ __launch_bounds__(256) void rowce_lse_kernel(const float* __restrict__ logits, int Br, int Bc, int off, int use_dcl,
                                                        float* __restrict__ lse) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Br) return;
    const float* lr = logits + (size_t)row * Bc;
    const int tgt = row + off;
    float m = -FLT_MAX, s = 0.f;
    for (int j = lane; j < Bc; j += 64) {
        if (use_dcl && j == tgt) continue;
        const float v = lr[j];
        const float mn = fmaxf(m, v);
        s = s * expf(m - mn) + expf(v - mn);
        m = mn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        const float mn = fmaxf(m, m2);
        s = s * expf(m - mn) + s2 * expf(m2 - mn);
        m = mn;
    }
    if (lane == 0) lse[row] = m + logf(s);
}

// loss_part = (1/Bg) * sum_i (lse[i] - logits[i, i+off]); one block, fixed summation order.
__global__ __launch_bounds__(1024) void rowce_sum_kernel(const float* __restrict__ logits, const float* __restrict__ lse, int Br, int Bc,
                                                         int off, float inv_bg, float* __restrict__ loss_part) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < Br; i += 1024) acc += lse[i] - logits[(size_t)i * Bc + i + off];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = 0.f;
        for (int k = 0; k < 16; ++k) v += red[k];
        loss_part[0] = v * inv_bg;
    }
}

__global__ __launch_bounds__(256) void rowce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ lse, int Br, int Bc,
                                                        int off, int use_dcl, const float* __restrict__ g, float gscale,
                                                        float* __restrict__ dlogits) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Br) return;
    const float gs = (g ? g[0] : 1.f) * gscale;
    const float* lr = logits + (size_t)row * Bc;
    float* dr = dlogits + (size_t)row * Bc;
    const float l = lse[row];
    const int tgt = row + off;
    for (int j = lane; j < Bc; j += 64) {
        float p = expf(lr[j] - l);
        if (j == tgt) p = use_dcl ? -1.f : p - 1.f;
        dr[j] = gs * p;
    }
}

static int rowce_check(const char* who, int Br, int Bc, int off, int Bg) {
    PCLIP_REQUIRE(Br >= 1 && Bc >= 1 && Bg >= Br && off >= 0 && off + Br <= Bc, "%s: Br=%d Bc=%d row_offset=%d Bg=%d", who, Br, Bc, off, Bg);
    return PCLIP_OK;
}

extern "C" int pclip_rowce_fwd(const float* logits, int Br, int Bc, int row_offset, int Bg, int use_dcl, float* lse, float* loss_part,
                               void* stream) {
    PCLIP_REQUIRE(logits && lse && loss_part, "rowce_fwd: null pointer");
    if (int rc = rowce_check("rowce_fwd", Br, Bc, row_offset, Bg)) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rowce_lse_kernel, dim3((Br + 3) / 4), dim3(256), 0, st, logits, Br, Bc, row_offset, use_dcl, lse);
    if (int rc = pclip_check_launch("rowce_lse_kernel")) return rc;
    hipLaunchKernelGGL(rowce_sum_kernel, dim3(1), dim3(1024), 0, st, logits, lse, Br, Bc, row_offset, 1.f / (float)Bg, loss_part);
    return pclip_check_launch("rowce_sum_kernel");
}

extern "C" int pclip_rowce_bwd(const float* logits, const float* lse, int Br, int Bc, int row_offset, int Bg, int use_dcl, const float* g,
                               float gscale, float* dlogits, void* stream) {
    PCLIP_REQUIRE(logits && lse && dlogits, "rowce_bwd: null pointer");
    if (int rc = rowce_check("rowce_bwd", Br, Bc, row_offset, Bg)) return rc;
    hipLaunchKernelGGL(rowce_bwd_kernel, dim3((Br + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, lse, Br, Bc, row_offset, use_dcl, g,
                       gscale / (float)Bg, dlogits);
    return pclip_check_launch("rowce_bwd_kernel");
}
