// Masked mean over the tokens of a sequence, forward and backward (f32, HBM-bound).
// Replaces Encoder._masked_mean (modules/models.py:76-90): sum_t h[b,t,:] * mask[b,t] divided by
// max(sum_t mask[b,t], 1) — the step that feeds the embedding head and, through it, the CLIP
// loss.  One workgroup per (sequence, 128-column slab): 8 token groups x 32 lanes x 4 columns,
// rows read as coalesced 16-byte pieces (eight per thread in flight), one pass over the tensor, partial
// sums met in LDS.
#include "pclip_common.h"

// grid (B, ceil(d / 128)); the mask byte is used by VALUE (the reference multiplies by the mask)
__global__ __launch_bounds__(256) void masked_mean_fwd_kernel(const float* __restrict__ h, const uint8_t* __restrict__ mask, int T, int d,
                                                              float* __restrict__ mean, float* __restrict__ inv_count) {
    __shared__ float part[8][128];
    __shared__ float cpart[8];
    const int b = blockIdx.x, k0 = blockIdx.y * 128;
    const int tg = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int k = k0 + 4 * l;
    const bool vec = (d & 3) == 0;
    const float* hb = h + (size_t)b * T * d;
    const uint8_t* mb = mask + (size_t)b * T;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, cnt = 0.f;
    if (vec && k + 3 < d) {
        // eight tokens per trip: their mask bytes first, then the rows of the valid ones, all in flight together
        for (int t0 = tg; t0 < T; t0 += 64) {
            float m[8];
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 8 * u;
                m[u] = t < T ? (float)mb[t] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m[u] != 0.f) v[u] = *reinterpret_cast<const float4*>(hb + (size_t)(t0 + 8 * u) * d + k);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cnt += m[u];
                a0 += m[u] * v[u].x; a1 += m[u] * v[u].y; a2 += m[u] * v[u].z; a3 += m[u] * v[u].w;
            }
        }
    } else {
        for (int t = tg; t < T; t += 8) {
            const float m = (float)mb[t];
            cnt += m;
            if (m == 0.f) continue;
            const float* row = hb + (size_t)t * d;
            if (k < d) a0 += m * row[k];
            if (k + 1 < d) a1 += m * row[k + 1];
            if (k + 2 < d) a2 += m * row[k + 2];
            if (k + 3 < d) a3 += m * row[k + 3];
        }
    }
    part[tg][4 * l] = a0; part[tg][4 * l + 1] = a1; part[tg][4 * l + 2] = a2; part[tg][4 * l + 3] = a3;
    if (l == 0) cpart[tg] = cnt;
    __syncthreads();
    if (threadIdx.x < 128) {
        float c = 0.f, s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) { c += cpart[g]; s += part[g][threadIdx.x]; }
        const float inv = 1.f / fmaxf(c, 1.f);
        if (k0 + threadIdx.x < d) mean[(size_t)b * d + k0 + threadIdx.x] = s * inv;
        if (threadIdx.x == 0 && blockIdx.y == 0) inv_count[b] = inv;
    }
}

// dh[b,t,:] = mask[b,t] * g[b,:] * inv_count[b].  A workgroup writes 32 token rows: 32 lanes x 16 bytes per row
// piece, eight rows per pass (the vector form needs d % 4 == 0; otherwise one wave per row, 4 bytes per lane).
__global__ __launch_bounds__(256) void masked_mean_bwd_kernel(const float* __restrict__ g, const uint8_t* __restrict__ mask,
                                                              const float* __restrict__ inv_count, int B, int T, int d,
                                                              float* __restrict__ dh) {
    const size_t nrows = (size_t)B * T;
    if ((d & 3) == 0) {
        const int l = threadIdx.x & 31, rg = threadIdx.x >> 5;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t row = (size_t)blockIdx.x * 32 + rg + 8 * u;
            if (row >= nrows) continue;
            const int b = (int)(row / T);
            const float w = (float)mask[row] * inv_count[b];
            const float4* gb = reinterpret_cast<const float4*>(g + (size_t)b * d);
            float4* out = reinterpret_cast<float4*>(dh + row * d);
            for (int k4 = l; k4 < (d >> 2); k4 += 32) {
                const float4 v = gb[k4];
                out[k4] = make_float4(w * v.x, w * v.y, w * v.z, w * v.w);
            }
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    for (int u = 0; u < 8; ++u) {
        const size_t row = (size_t)blockIdx.x * 32 + (threadIdx.x >> 6) + 4 * u;
        if (row >= nrows) continue;
        const int b = (int)(row / T);
        const float w = (float)mask[row] * inv_count[b];
        const float* gb = g + (size_t)b * d;
        float* out = dh + row * d;
        for (int k = lane; k < d; k += 64) out[k] = w * gb[k];
    }
}

extern "C" int pclip_masked_mean_fwd(const float* h, const uint8_t* mask, int B, int T, int d, float* mean, float* inv_count, void* stream) {
    PCLIP_REQUIRE(h && mask && mean && inv_count, "masked_mean_fwd: null pointer");
    PCLIP_REQUIRE(B >= 1 && B <= 65535 && T >= 1 && d >= 1, "masked_mean_fwd: B=%d T=%d d=%d", B, T, d);
    PCLIP_REQUIRE((d & 3) != 0 || (uintptr_t)h % 16 == 0, "masked_mean_fwd: h must be 16-byte aligned");
    hipLaunchKernelGGL(masked_mean_fwd_kernel, dim3(B, (d + 127) / 128), dim3(256), 0, (hipStream_t)stream, h, mask, T, d, mean, inv_count);
    return pclip_check_launch("masked_mean_fwd_kernel");
}

extern "C" int pclip_masked_mean_bwd(const float* g, const uint8_t* mask, const float* inv_count, int B, int T, int d, float* dh,
                                     void* stream) {
    PCLIP_REQUIRE(g && mask && inv_count && dh, "masked_mean_bwd: null pointer");
    PCLIP_REQUIRE(B >= 1 && T >= 1 && d >= 1, "masked_mean_bwd: B=%d T=%d d=%d", B, T, d);
    PCLIP_REQUIRE((d & 3) != 0 || ((uintptr_t)g | (uintptr_t)dh) % 16 == 0, "masked_mean_bwd: g and dh must be 16-byte aligned");
    const size_t rows = (size_t)B * T;
    hipLaunchKernelGGL(masked_mean_bwd_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, (hipStream_t)stream, g, mask, inv_count, B, T, d,
                       dh);
    return pclip_check_launch("masked_mean_bwd_kernel");
}
