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

// ---- both FILIP losses and their gradients in two launches ------------------------------
// (1/2) * (rowCE(sim_A) + rowCE(sim_B)) of modules/training_utils_filip.py:26-39 and its
// backward: one wave per row of either matrix computes the row's log-sum-exp, re-reads the
// row (L1/L2) to write d loss / d sim, and leaves lse - sim[i, i+off] for the summing block.
__global__ __launch_bounds__(256) void filip_loss_rows_kernel(const float* __restrict__ simA, const float* __restrict__ simB, int Br, int Bc,
                                                              int off, float gs, float* __restrict__ lse2, float* __restrict__ term2,
                                                              float* __restrict__ gA, float* __restrict__ gB) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Br) return;
    const int z = blockIdx.y;
    const float* lr = (z ? simB : simA) + (size_t)row * Bc;
    float* dr = (z ? gB : gA) + (size_t)row * Bc;
    const int tgt = row + off;
    float m = -FLT_MAX, s = 0.f;
    for (int j = lane; j < Bc; j += 64) {
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
    const float l = m + logf(s);
    if (lane == 0) {
        lse2[(size_t)z * Br + row] = l;
        term2[(size_t)z * Br + row] = l - lr[tgt];
    }
    for (int j = lane; j < Bc; j += 64) {
        float p = expf(lr[j] - l);
        if (j == tgt) p -= 1.f;
        dr[j] = gs * p;
    }
}

// loss3 = {(lossA + lossB) / 2, lossA, lossB}, loss? = (1/Bg) * sum of the row terms; one block, fixed order
__global__ __launch_bounds__(1024) void filip_loss_sum_kernel(const float* __restrict__ term2, int Br, float inv_bg, float* __restrict__ loss3) {
    __shared__ float red[2][16];
    float acc[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < Br; i += 1024) {
        acc[0] += term2[i];
        acc[1] += term2[(size_t)Br + i];
    }
    for (int z = 0; z < 2; ++z) {
        const float v = wave_sum(acc[z]);
        if ((threadIdx.x & 63) == 0) red[z][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < 16; ++k) { a += red[0][k]; b += red[1][k]; }
        a *= inv_bg; b *= inv_bg;
        loss3[0] = 0.5f * (a + b);
        loss3[1] = a;
        loss3[2] = b;
    }
}

extern "C" int pclip_filip_loss(const float* sim_A, const float* sim_B, int Br, int Bc, int row_offset, int Bg, float* lse2, float* loss3,
                                float* g_A, float* g_B, float* ws_terms, void* stream) {
    PCLIP_REQUIRE(sim_A && sim_B && lse2 && loss3 && g_A && g_B && ws_terms, "filip_loss: null pointer");
    if (int rc = rowce_check("filip_loss", Br, Bc, row_offset, Bg)) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(filip_loss_rows_kernel, dim3((Br + 3) / 4, 2), dim3(256), 0, st, sim_A, sim_B, Br, Bc, row_offset, 0.5f / (float)Bg, lse2,
                       ws_terms, g_A, g_B);
    if (int rc = pclip_check_launch("filip_loss_rows_kernel")) return rc;
    hipLaunchKernelGGL(filip_loss_sum_kernel, dim3(1), dim3(1024), 0, st, ws_terms, Br, 1.f / (float)Bg, loss3);
    return pclip_check_launch("filip_loss_sum_kernel");
}
