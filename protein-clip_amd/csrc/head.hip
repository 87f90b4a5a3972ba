// Embedding head of a CLIP tower after the pooling, forward and backward (f32, gfx950):
//   x_0 = pooled;  block l < nblk:  a = relu(W_l x_l + b_l),  y = LayerNorm(a) * gamma_l + beta_l,
//   x_{l+1} = y * keep_l * keep_scale;  z = W_nblk x_nblk + b_nblk;  out = z / max(|z|, 1e-12) * exp(t/2)
// (modules/models.py:49-51 with embedding_ffn as built at :54-73).  In PyTorch this is about ten
// launches forward and twenty backward over [B,128] tensors of 128 KB; here it is one launch forward
// and two backward.  Rows are independent except for the parameter gradients, so:
//   forward      one workgroup per 8 rows; activations stay in LDS between the layers; each layer's
//                weight matrix (64 KB, an L2 hit after the first workgroup) is staged in LDS with
//                coalesced loads, transposed so that thread o walks W[o][:] without bank conflicts;
//   backward 1   the same row split: d out -> d z -> ... -> d pooled, leaving d z and each block's
//                d (pre-activation) and d (LayerNorm output) in the workspace; weights staged as stored;
//   backward 2   one workgroup per 32 x 32 tile of a weight gradient G^T X on exact-f32 matrix cores
//                (v_mfma_f32_32x32x2_f32): both operands are read straight from memory as 128-byte row
//                segments, the rows are split over the four waves and the partial tiles met in LDS;
//                a few more workgroups sum the columns for d b, d gamma, d beta.
#include "pclip_common.h"

#define HEAD_MAXB PCLIP_HEAD_MAX_BLOCKS
#define HEAD_ROWS 8
#define HEAD_LD 132          // LDS row pitch of the activations (floats): rows stay 16-byte aligned
#define HEAD_WLD 129         // LDS row pitch of the staged weights: odd, so walking a column is conflict-free
#define HEAD_LDS_BYTES (128 * HEAD_WLD * 4)

struct HeadParams {
    const float* W[HEAD_MAXB + 1]; const float* b[HEAD_MAXB + 1];
    const float* gamma[HEAD_MAXB]; const float* beta[HEAD_MAXB];
};
struct HeadGrads {
    float* W[HEAD_MAXB + 1]; float* b[HEAD_MAXB + 1];
    float* gamma[HEAD_MAXB]; float* beta[HEAD_MAXB];
};

// saved activations: per block  a [B,d] | x_next [B,d] | mean [B] | rstd [B];  then z [B,d]
__host__ __device__ inline size_t head_save_block(int B, int d) { return 2 * (size_t)B * d + 2 * (size_t)B; }
// backward workspace: per block  d y [B,d] | d u [B,d];  then d z [B,d]
__host__ __device__ inline size_t head_ws_block(int B, int d) { return 2 * (size_t)B * d; }

__device__ __forceinline__ float head_sum16(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// W [d][d] row-major in memory -> registers (all 16-byte loads of a thread in flight together) -> LDS, as
// stored (wl[o * HEAD_WLD + k]) or transposed (wl[k * HEAD_WLD + o]).  The two halves are separate so that
// the next layer's loads can be in flight while this layer is computed.
__device__ __forceinline__ void head_load_weights(const float* __restrict__ W, float4 v[16], int d, int tid) {
    const int n4 = (d * d) >> 2;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int idx = tid + 256 * u;
        if (idx < n4) v[u] = reinterpret_cast<const float4*>(W)[idx];
    }
}
template <bool TRANSPOSE>
__device__ __forceinline__ void head_store_weights(const float4 v[16], float* __restrict__ wl, int d, int tid) {
    const int n4 = (d * d) >> 2;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int idx = tid + 256 * u;
        if (idx < n4) {
            const int o = (4 * idx) / d, k = (4 * idx) - o * d;
            if (TRANSPOSE) {
                wl[k * HEAD_WLD + o] = v[u].x; wl[(k + 1) * HEAD_WLD + o] = v[u].y;
                wl[(k + 2) * HEAD_WLD + o] = v[u].z; wl[(k + 3) * HEAD_WLD + o] = v[u].w;
            } else {
                wl[o * HEAD_WLD + k] = v[u].x; wl[o * HEAD_WLD + k + 1] = v[u].y;
                wl[o * HEAD_WLD + k + 2] = v[u].z; wl[o * HEAD_WLD + k + 3] = v[u].w;
            }
        }
    }
}

// acc[r] = sum_j wl[j * HEAD_WLD + c] v[4 rh + r][j]   for thread (c, rh): with the transposed weights this is
// (W v)[c], with the weights as stored it is (W^T v)[c].  Four steps of j per trip: their LDS reads are issued
// together (one wave per SIMD here, nothing else hides the latency).
__device__ __forceinline__ void head_matvec(const float* __restrict__ wl, const float (*vs)[HEAD_LD], int d, int c, int rh, float acc[4]) {
    if (c >= d) return;
#pragma unroll 4
    for (int j4 = 0; j4 < (d >> 2); ++j4) {
        const float w0 = wl[(4 * j4) * HEAD_WLD + c], w1 = wl[(4 * j4 + 1) * HEAD_WLD + c];
        const float w2 = wl[(4 * j4 + 2) * HEAD_WLD + c], w3 = wl[(4 * j4 + 3) * HEAD_WLD + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 x = *reinterpret_cast<const float4*>(&vs[4 * rh + r][4 * j4]);
            acc[r] += (w0 * x.x + w1 * x.y) + (w2 * x.z + w3 * x.w);
        }
    }
}

__global__ __launch_bounds__(256) void head_tail_fwd_kernel(const float* __restrict__ pooled, HeadParams P, const uint8_t* __restrict__ keep,
                                                            float keep_scale, float eps, const float* __restrict__ t, int B, int d, int nblk,
                                                            float* __restrict__ save, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float wl[];             // [128][HEAD_WLD] the layer's weights, transposed
    __shared__ __attribute__((aligned(16))) float xs[HEAD_ROWS][HEAD_LD];
    __shared__ __attribute__((aligned(16))) float ys[HEAD_ROWS][HEAD_LD];
    const int tid = threadIdx.x, row0 = blockIdx.x * HEAD_ROWS;
    const int o = tid & 127, rh = tid >> 7;
    const int qr = tid >> 4, ql = tid & 15;          // quarter-wave row work: row qr (threads 0-127), columns 4 ql.. and 64 + 4 ql..
    const size_t Bd = (size_t)B * d, sblk = head_save_block(B, d);
    float4 wv[16];
    head_load_weights(P.W[0], wv, d, tid);
    for (int idx = tid; idx < HEAD_ROWS * 128; idx += 256) {
        const int r = idx >> 7, k = idx & 127;
        xs[r][k] = (row0 + r < B && k < d) ? pooled[(size_t)(row0 + r) * d + k] : 0.f;
    }
    for (int l = 0; l <= nblk; ++l) {
        head_store_weights<true>(wv, wl, d, tid);
        __syncthreads();   // weights and the layer's input in place
        if (l < nblk) head_load_weights(P.W[l + 1], wv, d, tid);   // in flight during this layer
        const float b0 = o < d ? P.b[l][o] : 0.f;
        float acc[4] = {b0, b0, b0, b0};
        head_matvec(wl, xs, d, o, rh, acc);
        float* sv = l < nblk ? save + l * sblk : save + nblk * sblk;   // a_l or z
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = l < nblk ? fmaxf(acc[r], 0.f) : acc[r];
            ys[4 * rh + r][o] = o < d ? v : 0.f;
            if (o < d && row0 + 4 * rh + r < B) sv[(size_t)(row0 + 4 * rh + r) * d + o] = v;
        }
        __syncthreads();
        if (l == nblk) break;
        if (tid < 128) {   // LayerNorm + affine + dropout of row qr -> the next layer's input
            const int row = row0 + qr;
            const bool lo = 4 * ql < d, hi = 64 + 4 * ql < d;
            const float4 a0 = *reinterpret_cast<const float4*>(&ys[qr][4 * ql]);
            const float4 a1 = *reinterpret_cast<const float4*>(&ys[qr][64 + 4 * ql]);
            const float mean = head_sum16((a0.x + a0.y) + (a0.z + a0.w) + (a1.x + a1.y) + (a1.z + a1.w)) / (float)d;
            float c[8] = {a0.x - mean, a0.y - mean, a0.z - mean, a0.w - mean, a1.x - mean, a1.y - mean, a1.z - mean, a1.w - mean};
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += ((j < 4 ? lo : hi) ? c[j] * c[j] : 0.f);
            const float rstd = 1.f / sqrtf(head_sum16(sq) / (float)d + eps);
            float* xn = save + l * sblk + Bd;
            if (ql == 0 && row < B) { save[l * sblk + 2 * Bd + row] = mean; save[l * sblk + 2 * Bd + B + row] = rstd; }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int col = 64 * half + 4 * ql;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (col < d) {
                    const float4 g = *reinterpret_cast<const float4*>(P.gamma[l] + col), be = *reinterpret_cast<const float4*>(P.beta[l] + col);
                    v[0] = c[4 * half] * rstd * g.x + be.x; v[1] = c[4 * half + 1] * rstd * g.y + be.y;
                    v[2] = c[4 * half + 2] * rstd * g.z + be.z; v[3] = c[4 * half + 3] * rstd * g.w + be.w;
                    if (keep && row < B) {
                        const uchar4 kp = *reinterpret_cast<const uchar4*>(keep + (size_t)l * Bd + (size_t)row * d + col);
                        v[0] = kp.x ? v[0] * keep_scale : 0.f; v[1] = kp.y ? v[1] * keep_scale : 0.f;
                        v[2] = kp.z ? v[2] * keep_scale : 0.f; v[3] = kp.w ? v[3] * keep_scale : 0.f;
                    }
                    if (row < B) *reinterpret_cast<float4*>(xn + (size_t)row * d + col) = make_float4(v[0], v[1], v[2], v[3]);
                }
                *reinterpret_cast<float4*>(&xs[qr][col]) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
    }
    if (tid < 128) {   // out = z / max(|z|, 1e-12) * exp(t / 2)       (models.py:50-51)
        const int row = row0 + qr;
        const float4 z0 = *reinterpret_cast<const float4*>(&ys[qr][4 * ql]);
        const float4 z1 = *reinterpret_cast<const float4*>(&ys[qr][64 + 4 * ql]);
        const float ss = head_sum16(((z0.x * z0.x + z0.y * z0.y) + (z0.z * z0.z + z0.w * z0.w)) + ((z1.x * z1.x + z1.y * z1.y) + (z1.z * z1.z + z1.w * z1.w)));
        const float s = expf(t[0] * 0.5f), den = fmaxf(sqrtf(ss), 1e-12f);
        if (row < B) {
            if (4 * ql < d) *reinterpret_cast<float4*>(out + (size_t)row * d + 4 * ql) = make_float4(z0.x / den * s, z0.y / den * s, z0.z / den * s, z0.w / den * s);
            if (64 + 4 * ql < d)
                *reinterpret_cast<float4*>(out + (size_t)row * d + 64 + 4 * ql) = make_float4(z1.x / den * s, z1.y / den * s, z1.z / den * s, z1.w / den * s);
        }
    }
}

// Row-parallel half of the backward: d out -> d z -> (per block, last to first) d y, d u -> d pooled.
__global__ __launch_bounds__(256) void head_tail_bwd_rows_kernel(const float* __restrict__ dout, HeadParams P, const uint8_t* __restrict__ keep,
                                                                 float keep_scale, const float* __restrict__ t, int B, int d, int nblk,
                                                                 const float* __restrict__ save, float* __restrict__ ws,
                                                                 float* __restrict__ dpooled, float* __restrict__ dt) {
    extern __shared__ __attribute__((aligned(16))) float wl[];               // [128][HEAD_WLD] the layer's weights, as stored
    __shared__ __attribute__((aligned(16))) float gs[HEAD_ROWS][HEAD_LD];    // d (pre-activation of the layer being crossed)
    __shared__ __attribute__((aligned(16))) float dys[HEAD_ROWS][HEAD_LD];   // d (LayerNorm output) of the block below it
    __shared__ float dt_part[HEAD_ROWS];
    const int tid = threadIdx.x, row0 = blockIdx.x * HEAD_ROWS;
    const int k = tid & 127, rh = tid >> 7;
    const int qr = tid >> 4, ql = tid & 15;
    const size_t Bd = (size_t)B * d, sblk = head_save_block(B, d), wblk = head_ws_block(B, d);
    float4 wv[16];
    head_load_weights(P.W[nblk], wv, d, tid);
    if (tid < 128) {   // through the normalise: d z = (s / |z|) (d out - n (n . d out)),  dt += 0.5 s (n . d out)
        const int row = row0 + qr;
        const bool lo = 4 * ql < d && row < B, hi = 64 + 4 * ql < d && row < B;
        const float* zr = save + nblk * sblk + (size_t)row * d;
        const float* gr = dout + (size_t)row * d;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 z0 = lo ? *reinterpret_cast<const float4*>(zr + 4 * ql) : zero, z1 = hi ? *reinterpret_cast<const float4*>(zr + 64 + 4 * ql) : zero;
        const float4 g0 = lo ? *reinterpret_cast<const float4*>(gr + 4 * ql) : zero, g1 = hi ? *reinterpret_cast<const float4*>(gr + 64 + 4 * ql) : zero;
        const float ss = head_sum16(((z0.x * z0.x + z0.y * z0.y) + (z0.z * z0.z + z0.w * z0.w)) + ((z1.x * z1.x + z1.y * z1.y) + (z1.z * z1.z + z1.w * z1.w)));
        const float dot = head_sum16(((z0.x * g0.x + z0.y * g0.y) + (z0.z * g0.z + z0.w * g0.w)) + ((z1.x * g1.x + z1.y * g1.y) + (z1.z * g1.z + z1.w * g1.w)));
        const float s = expf(t[0] * 0.5f), nrm = sqrtf(ss);
        float f, c;   // d z = f (d out - c z)
        if (nrm >= 1e-12f) { const float inv = 1.f / nrm; f = s * inv; c = dot * inv * inv; if (ql == 0) dt_part[qr] = 0.5f * s * dot * inv; }
        else { f = s / 1e-12f; c = 0.f; if (ql == 0) dt_part[qr] = 0.5f * f * dot; }
        const float4 d0 = make_float4(f * (g0.x - c * z0.x), f * (g0.y - c * z0.y), f * (g0.z - c * z0.z), f * (g0.w - c * z0.w));
        const float4 d1 = make_float4(f * (g1.x - c * z1.x), f * (g1.y - c * z1.y), f * (g1.z - c * z1.z), f * (g1.w - c * z1.w));
        *reinterpret_cast<float4*>(&gs[qr][4 * ql]) = d0;
        *reinterpret_cast<float4*>(&gs[qr][64 + 4 * ql]) = d1;
        float* dz = ws + nblk * wblk + (size_t)row * d;
        if (lo) *reinterpret_cast<float4*>(dz + 4 * ql) = d0;
        if (hi) *reinterpret_cast<float4*>(dz + 64 + 4 * ql) = d1;
    }
    __syncthreads();
    if (tid == 0) {
        float v = 0.f;
        for (int r = 0; r < HEAD_ROWS; ++r) v += dt_part[r];
        if (v != 0.f) atomicAdd(dt, v);
    }
    for (int l = nblk; l >= 0; --l) {
        head_store_weights<false>(wv, wl, d, tid);
        __syncthreads();
        if (l > 0) head_load_weights(P.W[l - 1], wv, d, tid);   // in flight during this layer
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        head_matvec(wl, gs, d, k, rh, acc);   // d x_l = W_l^T d u_l
        if (l == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k < d && row0 + 4 * rh + r < B) dpooled[(size_t)(row0 + 4 * rh + r) * d + k] = acc[r];
            break;
        }
        const int bl = l - 1;   // the block whose output x_l is
        float* dy = ws + bl * wblk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * rh + r;
            float v = 0.f;
            if (k < d && row < B) {
                v = acc[r];
                if (keep) v = keep[(size_t)bl * Bd + (size_t)row * d + k] ? v * keep_scale : 0.f;
                dy[(size_t)row * d + k] = v;
            }
            dys[4 * rh + r][k] = v;
        }
        __syncthreads();
        if (tid < 128) {   // LayerNorm and ReLU backward of row qr
            const int row = row0 + qr;
            const float* ar = save + bl * sblk + (size_t)row * d;
            const float mean = row < B ? save[bl * sblk + 2 * Bd + row] : 0.f, rstd = row < B ? save[bl * sblk + 2 * Bd + B + row] : 0.f;
            float a[8], gy[8], xh[8];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int col = 64 * half + 4 * ql;
                const bool on = col < d && row < B;
                const float4 av = on ? *reinterpret_cast<const float4*>(ar + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 gm = on ? *reinterpret_cast<const float4*>(P.gamma[bl] + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 dv = *reinterpret_cast<const float4*>(&dys[qr][col]);
                a[4 * half] = av.x; a[4 * half + 1] = av.y; a[4 * half + 2] = av.z; a[4 * half + 3] = av.w;
                gy[4 * half] = dv.x * gm.x; gy[4 * half + 1] = dv.y * gm.y; gy[4 * half + 2] = dv.z * gm.z; gy[4 * half + 3] = dv.w * gm.w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[4 * half + j] = on ? (a[4 * half + j] - mean) * rstd : 0.f;
                    s1 += gy[4 * half + j];
                    s2 += gy[4 * half + j] * xh[4 * half + j];
                }
            }
            const float m1 = head_sum16(s1) / (float)d, m2 = head_sum16(s2) / (float)d;
            float* du = ws + bl * wblk + Bd + (size_t)row * d;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int col = 64 * half + 4 * ql;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float da = rstd * (gy[4 * half + j] - m1 - xh[4 * half + j] * m2);
                    v[j] = (col < d && row < B && a[4 * half + j] > 0.f) ? da : 0.f;
                }
                *reinterpret_cast<float4*>(&gs[qr][col]) = make_float4(v[0], v[1], v[2], v[3]);
                if (col < d && row < B) *reinterpret_cast<float4*>(du + col) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
    }
}

// Parameter half of the backward.  The first (nblk + 1) nt^2 workgroups (nt = ceil(d / 32)) each own a 32 x 32 tile
//   d W_m[o][k] = sum_r G_m[r][o] X_m[r][k]   with G_m = d z (m = nblk) or d u_m, X_m = pooled (m = 0) or x_m:
// wave w takes the rows r = 128 i + 8 u + 2 w + h; lane (c, h) of the 32x32x2 matrix instruction feeds G[r][o0 + c]
// and X[r][k0 + c], two 128-byte row segments per operand and instruction, sixteen instructions' loads in flight.
// The last (nblk + 1) nt workgroups sum columns: d b_m[o] = sum_r G_m[r][o], and for a block
// d gamma_m[o] = sum_r d y_m[r][o] xhat_m[r][o], d beta_m[o] = sum_r d y_m[r][o]   (8 row lanes x 32 columns).
__global__ __launch_bounds__(256) void head_tail_bwd_params_kernel(const float* __restrict__ pooled, int B, int d, int nblk,
                                                                   const float* __restrict__ save, const float* __restrict__ ws, HeadGrads Gd) {
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = (d + 31) / 32, ntiles = (nblk + 1) * nt * nt;
    const size_t Bd = (size_t)B * d, sblk = head_save_block(B, d), wblk = head_ws_block(B, d);
    if ((int)blockIdx.x >= ntiles) {   // column sums
        const int m = ((int)blockIdx.x - ntiles) / nt, o0 = (((int)blockIdx.x - ntiles) % nt) * 32;
        const int c = tid & 31, rr = tid >> 5;
        const bool norm = m < nblk, on = o0 + c < d;
        const float* G = (m == nblk ? ws + nblk * wblk : ws + m * wblk + Bd) + o0 + c;
        const float* dy = ws + (norm ? m : 0) * wblk + o0 + c;
        const float* av = save + (norm ? m : 0) * sblk + o0 + c;
        const float* mean = save + (norm ? m : 0) * sblk + 2 * Bd;
        const float* rstd = mean + B;
        float bsum = 0.f, gsum = 0.f, besum = 0.f;
        for (int r0 = rr; r0 < B; r0 += 64) {   // eight rows per trip, their loads issued together
            float g[8], dv[8], a[8], mu[8], rs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 8 * u;
                const bool ok = on && r < B;
                g[u] = ok ? G[(size_t)r * d] : 0.f;
                dv[u] = (ok && norm) ? dy[(size_t)r * d] : 0.f;
                a[u] = (ok && norm) ? av[(size_t)r * d] : 0.f;
                mu[u] = (ok && norm) ? mean[r] : 0.f;
                rs[u] = (ok && norm) ? rstd[r] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { bsum += g[u]; gsum += dv[u] * (a[u] - mu[u]) * rs[u]; besum += dv[u]; }
        }
        float (*part)[8][32] = reinterpret_cast<float (*)[8][32]>(&red[0][0][0]);
        part[0][rr][c] = bsum; part[1][rr][c] = gsum; part[2][rr][c] = besum;
        __syncthreads();
        if (tid < 32 && o0 + tid < d) {
            float v[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                v[q] = ((part[q][0][tid] + part[q][1][tid]) + (part[q][2][tid] + part[q][3][tid])) +
                       ((part[q][4][tid] + part[q][5][tid]) + (part[q][6][tid] + part[q][7][tid]));
            Gd.b[m][o0 + tid] = v[0];
            if (norm) { Gd.gamma[m][o0 + tid] = v[1]; Gd.beta[m][o0 + tid] = v[2]; }
        }
        return;
    }
    const int c = lane & 31, h = lane >> 5;
    const int m = blockIdx.x / (nt * nt), tile = blockIdx.x % (nt * nt);
    const int o0 = (tile / nt) * 32, k0 = (tile % nt) * 32;
    const float* G = (m == nblk ? ws + nblk * wblk : ws + m * wblk + Bd) + o0 + c;
    const float* X = (m == 0 ? pooled : save + (m - 1) * sblk + Bd) + k0 + c;
    const bool oon = o0 + c < d, kon = k0 + c < d;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r0 = 2 * w + h; r0 < B; r0 += 128) {
        float g[16], x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = r0 + 8 * u;
            g[u] = (oon && r < B) ? G[(size_t)r * d] : 0.f;
            x[u] = (kon && r < B) ? X[(size_t)r * d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g[u], x[u], acc, 0, 0, 0);
    }
    // the four waves' partial tiles: register q of lane (c, h) is (o0 + acc_row(q, h), k0 + c)
#pragma unroll
    for (int q = 0; q < 16; ++q) red[w][q][lane] = acc[q];
    __syncthreads();
    for (int idx = tid; idx < 1024; idx += 256) {   // element (q, lane) of the tile, one output row per 32 threads
        const int q = idx >> 6, ln = idx & 63;
        const int o = o0 + acc_row(q, ln >> 5), k = k0 + (ln & 31);
        const float v = (red[0][q][ln] + red[1][q][ln]) + (red[2][q][ln] + red[3][q][ln]);
        if (o < d && k < d) Gd.W[m][(size_t)o * d + k] = v;
    }
}

// ---- host side -------------------------------------------------------------------------
static PclipLdsLimit g_head_fwd_lds, g_head_bwd_lds;

static int head_check(const char* who, int B, int d, int nblk) {
    PCLIP_REQUIRE(B >= 1 && B <= (1 << 24), "%s: B=%d out of range", who, B);
    PCLIP_REQUIRE(d >= 4 && d % 4 == 0 && d <= 128, "%s: d=%d (need a multiple of 4, <= 128)", who, d);
    PCLIP_REQUIRE(nblk >= 0 && nblk <= HEAD_MAXB, "%s: %d blocks (at most %d)", who, nblk, HEAD_MAXB);
    return PCLIP_OK;
}

extern "C" size_t pclip_head_tail_save_floats(int B, int d, int nblk) {
    if (B < 1 || d < 1 || nblk < 0) return 0;
    return nblk * head_save_block(B, d) + (size_t)B * d;
}

extern "C" size_t pclip_head_tail_bwd_workspace_bytes(int B, int d, int nblk) {
    if (B < 1 || d < 1 || nblk < 0) return 0;
    return (nblk * head_ws_block(B, d) + (size_t)B * d) * sizeof(float);
}

extern "C" int pclip_head_tail_fwd(const float* pooled, const float* const* params, int nblk, const uint8_t* keep, float keep_scale, float ln_eps,
                                   const float* t, int B, int d, float* save, float* out, void* stream) {
    if (int rc = head_check("head_tail_fwd", B, d, nblk)) return rc;
    PCLIP_REQUIRE(pooled && params && t && save && out, "head_tail_fwd: null pointer");
    HeadParams P = {};
    uintptr_t al = (uintptr_t)pooled | (uintptr_t)save | (uintptr_t)out | (uintptr_t)keep;
    for (int l = 0; l <= nblk; ++l) {
        const int base = 4 * l;
        P.W[l] = params[base]; P.b[l] = params[base + 1];
        PCLIP_REQUIRE(P.W[l] && P.b[l], "head_tail_fwd: null parameter of layer %d", l);
        al |= (uintptr_t)P.W[l];
        if (l < nblk) {
            P.gamma[l] = params[base + 2]; P.beta[l] = params[base + 3];
            PCLIP_REQUIRE(P.gamma[l] && P.beta[l], "head_tail_fwd: null LayerNorm parameter of block %d", l);
            al |= (uintptr_t)P.gamma[l] | (uintptr_t)P.beta[l];
        }
    }
    PCLIP_REQUIRE(al % 16 == 0 && ((uintptr_t)keep % 4 == 0), "head_tail_fwd: tensors must be 16-byte aligned");
    if (int rc = pclip_ensure_lds(g_head_fwd_lds, (const void*)head_tail_fwd_kernel, HEAD_LDS_BYTES, "head_tail_fwd")) return rc;
    hipLaunchKernelGGL(head_tail_fwd_kernel, dim3((B + HEAD_ROWS - 1) / HEAD_ROWS), dim3(256), HEAD_LDS_BYTES, (hipStream_t)stream, pooled, P, keep, keep_scale,
                       ln_eps, t, B, d, nblk, save, out);
    return pclip_check_launch("head_tail_fwd_kernel");
}

extern "C" int pclip_head_tail_bwd(const float* pooled, const float* const* params, int nblk, const uint8_t* keep, float keep_scale, const float* t,
                                   const float* save, const float* dout, int B, int d, float* dpooled, float* const* dparams, float* dt, void* ws,
                                   size_t ws_bytes, void* stream) {
    if (int rc = head_check("head_tail_bwd", B, d, nblk)) return rc;
    PCLIP_REQUIRE(pooled && params && t && save && dout && dpooled && dparams && dt && ws, "head_tail_bwd: null pointer");
    PCLIP_REQUIRE(ws_bytes >= pclip_head_tail_bwd_workspace_bytes(B, d, nblk), "head_tail_bwd: workspace of %zu bytes, need %zu", ws_bytes,
                  pclip_head_tail_bwd_workspace_bytes(B, d, nblk));
    HeadParams P = {};
    HeadGrads Gd = {};
    uintptr_t al = (uintptr_t)pooled | (uintptr_t)save | (uintptr_t)dout | (uintptr_t)dpooled | (uintptr_t)ws;
    for (int l = 0; l <= nblk; ++l) {
        const int base = 4 * l;
        P.W[l] = params[base]; P.b[l] = params[base + 1];
        Gd.W[l] = dparams[base]; Gd.b[l] = dparams[base + 1];
        PCLIP_REQUIRE(P.W[l] && P.b[l] && Gd.W[l] && Gd.b[l], "head_tail_bwd: null parameter or gradient of layer %d", l);
        al |= (uintptr_t)P.W[l] | (uintptr_t)Gd.W[l];
        if (l < nblk) {
            P.gamma[l] = params[base + 2]; P.beta[l] = params[base + 3];
            Gd.gamma[l] = dparams[base + 2]; Gd.beta[l] = dparams[base + 3];
            PCLIP_REQUIRE(P.gamma[l] && P.beta[l] && Gd.gamma[l] && Gd.beta[l], "head_tail_bwd: null LayerNorm parameter or gradient of block %d", l);
            al |= (uintptr_t)P.gamma[l];
        }
    }
    PCLIP_REQUIRE(al % 16 == 0, "head_tail_bwd: tensors must be 16-byte aligned");
    if (int rc = pclip_ensure_lds(g_head_bwd_lds, (const void*)head_tail_bwd_rows_kernel, HEAD_LDS_BYTES, "head_tail_bwd")) return rc;
    hipLaunchKernelGGL(head_tail_bwd_rows_kernel, dim3((B + HEAD_ROWS - 1) / HEAD_ROWS), dim3(256), HEAD_LDS_BYTES, (hipStream_t)stream, dout, P, keep, keep_scale,
                       t, B, d, nblk, save, (float*)ws, dpooled, dt);
    if (int rc = pclip_check_launch("head_tail_bwd_rows_kernel")) return rc;
    const int nt = (d + 31) / 32;
    hipLaunchKernelGGL(head_tail_bwd_params_kernel, dim3((nblk + 1) * (nt * nt + nt)), dim3(256), 0, (hipStream_t)stream, pooled, B, d, nblk, save,
                       (const float*)ws, Gd);
    return pclip_check_launch("head_tail_bwd_params_kernel");
}
