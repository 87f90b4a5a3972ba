// CLIP symmetric InfoNCE for tall row blocks (Br >= 128; one rank of a large global batch, or one GPU with
// B > 512), exact-f32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.  Same mathematics and argument
// blocks as clip.hip; what differs is where the operands live:
//   * a workgroup's four waves own four DIFFERENT 32-row tiles of A (128 rows), whose fragments stay in
//     registers for the whole kernel (64 per lane: lane half h holds k = 64 h .. 64 h + 63 of its row);
//   * the 32-row tiles of Bm are read from memory ONCE per workgroup with coalesced 16-byte loads (a row
//     is 512 contiguous bytes), double-buffered in LDS and shared by the four waves; the loads of tile
//     n + 1 are in flight while tile n is multiplied; one barrier per tile (4096 matrix-core cycles);
//   * LDS rows have a pitch of 136 floats and the 16-byte column groups of rows 8..15, 24..31 are
//     swapped in pairs, so that both access patterns are conflict-free: a lane reading its own row 16 bytes
//     at a time (first product) and 32 lanes reading 32 consecutive floats of one row (second product).
// The contraction order is k = 64 h + s for lane half h at step s, the same for both operands.
#include "clip_args.h"

#define CT_LD 136
#define CT_TILE (32 * CT_LD + 32)      // floats per ring slot: the tile, then the 32 log-sum-exps of its rows (backward)
#define CT_RING_BYTES (2 * CT_TILE * 4)
#define CT_OUT_BYTES (4 * 32 * 129 * 4)

struct TileStage {
    float4 v[4];
    float lb;
};

// rows j0 .. j0 + 31 of Bm (clamped to the last row), 128 columns (zero beyond d): thread -> (row, 16-byte group)
__device__ __forceinline__ void ct_load_tile(TileStage& st, const ClipProb& P, int d, int j0, int tid, bool want_lse) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + 256 * u, row = idx >> 5, k4 = idx & 31;
        const float* src = P.Bm + (size_t)min(j0 + row, P.Bc - 1) * d + 4 * k4;
        st.v[u] = 4 * k4 < d ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (want_lse && tid < 32) st.lb = P.lseB[min(j0 + tid, P.Bc - 1)];
}
__device__ __forceinline__ void ct_store_tile(const TileStage& st, float* slot, int tid, bool want_lse) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + 256 * u, row = idx >> 5, k4 = idx & 31;
        *reinterpret_cast<float4*>(slot + row * CT_LD + 4 * (k4 ^ ((row >> 3) & 1))) = st.v[u];
    }
    if (want_lse && tid < 32) slot[32 * CT_LD + tid] = st.lb;
}

// the caller's row as matrix-core fragments: afr[m] = A[row][64 h + 4 m .. + 3]
__device__ __forceinline__ void ct_load_a(float4 (&afr)[16], const float* __restrict__ arow, int d, int h) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int k = 64 * h + 4 * m;
        afr[m] = k < d ? *reinterpret_cast<const float4*>(arow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// T[j][i] = Bm_j . A_i for the staged tile: lane (c, h) reads row c of the slot, 16 bytes at a time
__device__ __forceinline__ f32x16 ct_logit_tile(const float4 (&afr)[16], const float* slot, int c, int h) {
    const float* tb = slot + c * CT_LD + 64 * h;
    const int sw = (c >> 3) & 1;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const float4 b = *reinterpret_cast<const float4*>(tb + 4 * (m ^ sw));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, afr[m].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, afr[m].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, afr[m].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, afr[m].w, acc, 0, 0, 0);
    }
    return acc;
}

// grid (ceil(Br / 128), column splits, directions).  Every lane keeps a running (max, sum) of its logit row.
__global__ __launch_bounds__(256) void clip_lse_tiled_kernel(ClipArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ClipProb& P = args.p[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int d = args.d;
    const int i0 = blockIdx.x * 128 + 32 * w;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) *args.fin_counter = 0u;   // for the finalize launch behind this one
    const bool wave_on = i0 < P.Br;
    const int irow = i0 + c;
    const bool ivalid = irow < P.Br;
    float4 afr[16];
    ct_load_a(afr, P.A + (size_t)min(irow, P.Br - 1) * d, d, h);
    const int nct = (P.Bc + 31) >> 5;
    const int ct_begin = blockIdx.y * args.tiles_per_split;
    const int ct_end = min(nct, ct_begin + args.tiles_per_split);
    TileStage st;
    if (ct_begin < ct_end) {
        ct_load_tile(st, P, d, ct_begin * 32, tid, false);
        ct_store_tile(st, lds, tid, false);
    }
    __syncthreads();
    float m = -FLT_MAX, s = 0.f;
    for (int ct = ct_begin, n = 0; ct < ct_end; ++ct, ++n) {
        const bool more = ct + 1 < ct_end;
        if (more) ct_load_tile(st, P, d, (ct + 1) * 32, tid, false);
        if (wave_on) {
            const int j0 = ct * 32;
            f32x16 acc = ct_logit_tile(afr, lds + (n & 1) * CT_TILE, c, h);
            // the diagonal element S[i, i+off] if it falls in this tile (wave-uniform band test)
            if (j0 < i0 + P.off + 32 && j0 + 32 > i0 + P.off) {
                const int tj = irow + P.off - j0;
                if (tj >= 0 && tj < 32 && ((tj >> 2) & 1) == h && ivalid) {
                    const int rr = (tj & 3) + 4 * (tj >> 3);
                    float dv = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dv = (r == rr) ? acc[r] : dv;
                    P.diag[irow] = dv;
                }
            }
            if (P.logits && ivalid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + acc_row(r, h);
                    if (j < P.Bc) P.logits[(size_t)irow * P.Bc + j] = (args.logits_mode == 2) ? expf(acc[r]) : acc[r];
                }
            }
            if (j0 + 32 > P.Bc) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (j0 + acc_row(r, h) >= P.Bc) acc[r] = -FLT_MAX;
            }
            float tm = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tm = fmaxf(tm, acc[r]);
            const float mn = fmaxf(m, tm);
            float ts = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ts += expf(acc[r] - mn);
            s = s * expf(m - mn) + ts;
            m = mn;
        }
        if (more) ct_store_tile(st, lds + ((n + 1) & 1) * CT_TILE, tid, false);
        __syncthreads();
    }
    const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
    const float M = fmaxf(m, m2);
    const float S = s * expf(m - M) + s2 * expf(m2 - M);
    if (h == 0 && ivalid) P.part[(size_t)blockIdx.y * P.Br + irow] = make_float2(M, S);
}

// Backward: per tile recompute T, form G^T[j][i] = scale (exp(T - lseA_i) + exp(T - lseB_j)) - 2 scale [j == i + off], then
// dA^T[dcol][i] += sum_j Bm[j][dcol] G^T[j][i] with G^T's registers as the B operand and Bm read from the staged tile
// (32 consecutive floats of row acc_row(q, h) per lane half).  A wave's 32 x 128 result leaves through LDS as whole rows.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void clip_grad_tiled_kernel(ClipArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ClipProb& P = args.p[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int d = args.d;
    const int i0 = blockIdx.x * 128 + 32 * w;
    const bool wave_on = i0 < P.Br;
    const int irow = i0 + c;
    float4 afr[16];
    ct_load_a(afr, P.A + (size_t)min(irow, P.Br - 1) * d, d, h);
    const float lseA = P.lseA[min(irow, P.Br - 1)];
    const float scale = args.scale * (args.g ? args.g[0] : 1.f);
    const int nct = (P.Bc + 31) >> 5;
    const int ct_begin = blockIdx.y * args.tiles_per_split;
    const int ct_end = min(nct, ct_begin + args.tiles_per_split);
    f32x16 dacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dacc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    TileStage st;
    if (ct_begin < ct_end) {
        ct_load_tile(st, P, d, ct_begin * 32, tid, true);
        ct_store_tile(st, lds, tid, true);
    }
    __syncthreads();
    for (int ct = ct_begin, n = 0; ct < ct_end; ++ct, ++n) {
        const bool more = ct + 1 < ct_end;
        if (more) ct_load_tile(st, P, d, (ct + 1) * 32, tid, true);
        if (wave_on) {
            const int j0 = ct * 32;
            const float* slot = lds + (n & 1) * CT_TILE;
            f32x16 acc = ct_logit_tile(afr, slot, c, h);
            const int tdiag = irow + P.off - j0;  // local row holding the target of this lane's i
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int jl = acc_row(q, h);
                const float lb = slot[32 * CT_LD + jl];
                float gq = scale * (expf(acc[q] - lseA) + expf(acc[q] - lb));
                if (jl == tdiag) gq -= 2.f * scale;
                acc[q] = (j0 + jl < P.Bc) ? gq : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = acc_row(q, h);
                const float* br = slot + row * CT_LD + (c & 3);
                const int rsw = (row >> 3) & 1;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float bv = br[4 * (((32 * t + c) >> 2) ^ rsw)];
                    dacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, acc[q], dacc[t], 0, 0, 0);
                }
            }
        }
        if (more) ct_store_tile(st, lds + ((n + 1) & 1) * CT_TILE, tid, true);
        __syncthreads();
    }
    // register q of dacc[t] in lane (c, h) is (column 32 t + acc_row(q, h), row i0 + c): through this wave's part of LDS
    float* mine = lds + (size_t)w * 32 * 129;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) mine[c * 129 + 32 * t + acc_row(q, h)] = dacc[t][q];
    __builtin_amdgcn_wave_barrier();
    if (wave_on) {
        float* out = P.slab + ((size_t)blockIdx.y * P.Br + i0) * d;
        for (int idx = lane; idx < 32 * d; idx += 64) {
            const int i = idx / d, k = idx - i * d;
            if (i0 + i < P.Br) out[(size_t)i * d + k] = mine[i * 129 + k];
        }
    }
}

// ---- host side -------------------------------------------------------------------------
size_t clip_tiled_lds_bytes(bool backward) { return backward ? (CT_OUT_BYTES > CT_RING_BYTES ? CT_OUT_BYTES : CT_RING_BYTES) : CT_RING_BYTES; }

int clip_lse_tiled_launch(const ClipArgs& a, int Br, hipStream_t st) {
    hipLaunchKernelGGL(clip_lse_tiled_kernel, dim3((Br + 127) / 128, a.nsplit, 2), dim3(256), CT_RING_BYTES, st, a);
    return pclip_check_launch("clip_lse_tiled_kernel");
}

int clip_grad_tiled_launch(const ClipArgs& a, int Br, hipStream_t st) {
    static PclipLdsLimit lds_limit;   // 66 KB of dynamic LDS is above the default launch limit
    const size_t bytes = clip_tiled_lds_bytes(true);
    if (int rc = pclip_ensure_lds(lds_limit, (const void*)clip_grad_tiled_kernel, bytes, "clip_grad_tiled_kernel")) return rc;
    hipLaunchKernelGGL(clip_grad_tiled_kernel, dim3((Br + 127) / 128, a.nsplit, 2), dim3(256), bytes, st, a);
    return pclip_check_launch("clip_grad_tiled_kernel");
}
