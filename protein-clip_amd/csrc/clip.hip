// CLIP symmetric InfoNCE on exact-f32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Both directions of the loss are the same problem: for the caller's rows A (X or Y) against
// the whole other side Bm (Y or X), a max-shifted log-sum-exp over the columns.  A wave
// computes a TRANSPOSED 32x32 logit tile T[j][i] = Bm_j . A_i, so a row i of the logits
// lives in ONE lane: the soft-max statistics are reduced over registers, never across lanes,
// and in the backward the tile of softmax gradients is already laid out as the B operand of
// the second product dA^T = Bm^T G^T (no LDS round trip, no lane movement).
//
// f32 operands are loaded as float4 per lane; the k order inside the contraction is
// permuted identically for both operands (lane half h takes k = 8m+4h+q), which leaves the
// dot products unchanged.
#include "pclip_common.h"

#include "clip_args.h"

// T[j][i] for the 32 rows j0.. of Bm and the 32 rows i0.. of A.  Out-of-range rows are
// clamped to the last valid row (finite garbage, masked by the callers).
__device__ __forceinline__ f32x16 logit_tile_T(const float* __restrict__ arow, const float* __restrict__ brow, int d, int h) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = 0; k0 < d; k0 += 64) {
        float4 a4[8], b4[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int kk = k0 + 8 * m + 4 * h;
            if (kk < d) {
                a4[m] = *reinterpret_cast<const float4*>(arow + kk);
                b4[m] = *reinterpret_cast<const float4*>(brow + kk);
            } else {
                a4[m] = make_float4(0.f, 0.f, 0.f, 0.f);
                b4[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[m].x, a4[m].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[m].y, a4[m].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[m].z, a4[m].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[m].w, a4[m].w, acc, 0, 0, 0);
        }
    }
    return acc;
}

// grid (row tiles, column splits, directions); 4 waves per block, wave w takes column tiles
// ct_begin + w, +4, ...  Each lane keeps a running (max, sum) for its logit row.
__global__ __launch_bounds__(256) void clip_lse_kernel(ClipArgs args) {
    const ClipProb& P = args.p[blockIdx.z];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int d = args.d;
    const int i0 = blockIdx.x * 32;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *args.fin_counter = 0u;   // for the finalize launch behind this one
    if (i0 >= P.Br) return;
    const int irow = i0 + c;
    const bool ivalid = irow < P.Br;
    const float* arow = P.A + (size_t)min(irow, P.Br - 1) * d;
    const int nct = (P.Bc + 31) >> 5;
    const int ct_begin = blockIdx.y * args.tiles_per_split;
    const int ct_end = min(nct, ct_begin + args.tiles_per_split);
    float m = -FLT_MAX, s = 0.f;
    for (int ct = ct_begin + w; ct < ct_end; ct += 4) {
        const int j0 = ct * 32;
        const float* brow = P.Bm + (size_t)min(j0 + c, P.Bc - 1) * d;
        f32x16 acc = logit_tile_T(arow, brow, d, h);
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
    const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
    const float M = fmaxf(m, m2);
    const float S = s * expf(m - M) + s2 * expf(m2 - M);
    if (h == 0 && ivalid) P.part[(size_t)(blockIdx.y * 4 + w) * P.Br + irow] = make_float2(M, S);
}

// Merges the partial (max, sum) pairs into lse, one thread per (direction, row), and reduces the loss share
// scale * sum_i (lse_row_i + lse_col_i - 2 S_ii): every block leaves its partial sum, the block that arrives
// last (a ticket, cleared by the score kernel in front of this launch) adds them up in block order, so the
// result does not depend on the order in which the blocks ran.
#define CLIP_FIN_THREADS 128
__global__ __launch_bounds__(CLIP_FIN_THREADS) void clip_finalize_kernel(ClipArgs args, float* __restrict__ loss_part) {
    __shared__ float red[CLIP_FIN_THREADS / 64];
    __shared__ unsigned ticket;
    const int npart = args.npart, Br = args.p[0].Br;
    const int item = blockIdx.x * CLIP_FIN_THREADS + threadIdx.x;
    float acc = 0.f;
    if (item < 2 * Br) {
        const ClipProb& P = args.p[item >= Br];
        const int i = item >= Br ? item - Br : item;
        const float2* part = P.part + i;
        float M = -FLT_MAX, S = 0.f;
        for (int k0 = 0; k0 < npart; k0 += 8) {   // eight partials requested at once, merged in order
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k0 + u < npart ? part[(size_t)(k0 + u) * Br] : make_float2(-FLT_MAX, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float Mn = fmaxf(M, v[u].x);
                S = S * expf(M - Mn) + v[u].y * expf(v[u].x - Mn);
                M = Mn;
            }
        }
        const float lse = M + logf(S);
        P.lse_out[i] = lse;
        acc = lse - P.diag[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = 0.f;
        for (int k = 0; k < CLIP_FIN_THREADS / 64; ++k) v += red[k];
        args.fin_part[blockIdx.x] = v;
        __threadfence();   // the partial sum is visible before the ticket is
        ticket = atomicAdd(args.fin_counter, 1u);
    }
    __syncthreads();
    if (ticket == gridDim.x - 1 && threadIdx.x == 0) {
        __threadfence();
        float v = 0.f;
        for (unsigned k = 0; k < gridDim.x; ++k) v += __hip_atomic_load(args.fin_part + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        loss_part[0] = v * args.scale;
    }
}

// Backward: per tile recompute T, form G^T[j][i] = scale*(exp(T-lseA_i) + exp(T-lseB_j)) - 2 scale [j == i+off],
// then dA^T[dcol][i] += sum_j Bm[j][dcol] G^T[j][i] on the matrix cores with G^T's registers as
// the B operand (k pairs (row(q,0), row(q,1)) match the accumulator's lane halves).
__global__ __launch_bounds__(256) void clip_grad_kernel(ClipArgs args) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4][32][129]
    const ClipProb& P = args.p[blockIdx.z];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int d = args.d;
    const int i0 = blockIdx.x * 32;
    if (i0 >= P.Br) return;
    const int irow = i0 + c;
    const float* arow = P.A + (size_t)min(irow, P.Br - 1) * d;
    const float lseA = P.lseA[min(irow, P.Br - 1)];
    const float scale = args.scale * (args.g ? args.g[0] : 1.f);
    const int nct = (P.Bc + 31) >> 5;
    const int ct_begin = blockIdx.y * args.tiles_per_split;
    const int ct_end = min(nct, ct_begin + args.tiles_per_split);
    f32x16 dacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dacc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int ct = ct_begin + w; ct < ct_end; ct += 4) {
        const int j0 = ct * 32;
        const float* brow = P.Bm + (size_t)min(j0 + c, P.Bc - 1) * d;
        f32x16 acc = logit_tile_T(arow, brow, d, h);
        const int tdiag = irow + P.off - j0;  // local row holding the target of this lane's i
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int jl = acc_row(q, h);
            const int j = j0 + jl;
            const float lb = P.lseB[min(j, P.Bc - 1)];
            float gq = scale * (expf(acc[q] - lseA) + expf(acc[q] - lb));
            if (jl == tdiag) gq -= 2.f * scale;
            acc[q] = (j < P.Bc) ? gq : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float* bq = P.Bm + (size_t)min(j0 + acc_row(q, h), P.Bc - 1) * d;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int dc = 32 * t + c;
                const float bv = (dc < d) ? bq[dc] : 0.f;
                dacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, acc[q], dacc[t], 0, 0, 0);
            }
        }
    }
    // sum the four waves' partial dA^T through LDS and store this block's slab rows
    float* mine = red + (size_t)w * 32 * 129 + c * 129;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) mine[32 * t + acc_row(q, h)] = dacc[t][q];
    __syncthreads();
    float* out = P.slab + (size_t)blockIdx.y * P.Br * d;
    for (int idx = threadIdx.x; idx < 32 * d; idx += 256) {
        const int i = idx / d, k = idx - i * d;
        if (i0 + i < P.Br) {
            const float v = red[i * 129 + k] + red[(32 + i) * 129 + k] + red[(64 + i) * 129 + k] + red[(96 + i) * 129 + k];
            out[(size_t)(i0 + i) * d + k] = v;
        }
    }
}

// out[i] = sum over slabs, 16 bytes per thread; blockIdx.y = side (slabs of side 1 follow those of side 0)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float4* __restrict__ slab, int nslab, size_t stride4, float4* __restrict__ out0,
                                                       float4* __restrict__ out1, size_t n4) {
    const float4* src = slab + (size_t)blockIdx.y * nslab * stride4;
    float4* out = blockIdx.y ? out1 : out0;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < nslab; s0 += 8) {
            float4 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = s0 + u < nslab ? src[(size_t)(s0 + u) * stride4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { v.x += a[u].x; v.y += a[u].y; v.z += a[u].z; v.w += a[u].w; }
        }
        out[i] = v;
    }
}

// ---- host side -------------------------------------------------------------------------
// Which kernels and how many column splits.  Tall row blocks (one rank of a large batch) take the tiled kernels of
// clip_tiled.hip: 128 rows per workgroup, about 512 workgroups, at least four column tiles each.
struct ClipPlan { int nsplit, tiles_per_split, npart; bool tiled; };
static ClipPlan clip_plan(int Br, int Bg, bool backward) {
    ClipPlan pl;
    const int nct = (Bg + 31) / 32;
    pl.tiled = Br >= CLIP_TILED_MIN_ROWS;
    if (pl.tiled) {
        const int rowblocks = (Br + 127) / 128;
        // two workgroups per CU (measured at 1024 x 8192: 256 and 1024 workgroups are 4-7 % and 8-40 % slower)
        int ns = 512 / (2 * rowblocks);
        if (ns > nct / 4) ns = nct / 4;
        if (ns > 64) ns = 64;
        if (ns < 1) ns = 1;
        pl.tiles_per_split = (nct + ns - 1) / ns;
        pl.nsplit = (nct + pl.tiles_per_split - 1) / pl.tiles_per_split;
        pl.npart = pl.nsplit;
        return pl;
    }
    const int rowtiles = (Br + 31) / 32;
    int ns = (nct + 3) / 4;                       // at least one tile per wave
    int cap = 1024 / (rowtiles * 2);              // about 4 blocks per CU at most
    if (cap < 1) cap = 1;
    if (ns > cap) ns = cap;
    if (ns > 64) ns = 64;
    int tps = (nct + ns - 1) / ns;
    tps = (tps + 3) / 4 * 4;
    pl.tiles_per_split = tps;
    pl.nsplit = (nct + tps - 1) / tps;
    pl.npart = pl.nsplit * 4;
    return pl;
}

static int clip_fin_blocks(int Br) { return (2 * Br + CLIP_FIN_THREADS - 1) / CLIP_FIN_THREADS; }
static size_t clip_fwd_ws(int Br, int npart) {  // part[2][npart][Br] float2 | diag[2][Br] | ticket | partial losses of the finalize blocks
    return pclip_align_up((size_t)2 * npart * Br * sizeof(float2), 256) + pclip_align_up((size_t)2 * Br * sizeof(float), 256) + 256 +
           pclip_align_up((size_t)clip_fin_blocks(Br) * sizeof(float), 256);
}
static size_t clip_bwd_ws(int Br, int d, int nsplit) {  // slab[2][nsplit][Br][d]
    return pclip_align_up((size_t)2 * nsplit * Br * d * sizeof(float), 256);
}

extern "C" size_t pclip_clip_workspace_bytes(int Br, int Bg, int d) {
    if (Br < 1 || Bg < 1 || d < 1) return 0;
    size_t a = clip_fwd_ws(Br, clip_plan(Br, Bg, false).npart), b = clip_bwd_ws(Br, d, clip_plan(Br, Bg, true).nsplit);
    return a > b ? a : b;
}

static int clip_check(const char* who, int Br, int Bg, int d, int row_offset) {
    PCLIP_REQUIRE(Br >= 1 && Bg >= Br, "%s: Br=%d Bg=%d (need 1 <= Br <= Bg)", who, Br, Bg);
    PCLIP_REQUIRE(d >= 4 && d % 4 == 0 && d <= PCLIP_CLIP_MAX_D, "%s: d=%d (need a multiple of 4, <= %d)", who, d, PCLIP_CLIP_MAX_D);
    PCLIP_REQUIRE(row_offset >= 0 && row_offset + Br <= Bg, "%s: row_offset=%d with Br=%d Bg=%d", who, row_offset, Br, Bg);
    return PCLIP_OK;
}

extern "C" int pclip_clip_fwd(const float* X_own, const float* Y_own, const float* X_all, const float* Y_all, int Br, int Bg,
                              int d, int row_offset, float* lse_row, float* lse_col, float* loss_part, float* logits_out,
                              int logits_mode, void* ws, size_t ws_bytes, void* stream) {
    PCLIP_REQUIRE(X_own && Y_own && X_all && Y_all && lse_row && lse_col && loss_part && ws, "clip_fwd: null pointer");
    if (int rc = clip_check("clip_fwd", Br, Bg, d, row_offset)) return rc;
    PCLIP_REQUIRE(logits_mode >= 0 && logits_mode <= 2, "clip_fwd: logits_mode=%d", logits_mode);
    PCLIP_REQUIRE(((uintptr_t)X_own | (uintptr_t)Y_own | (uintptr_t)X_all | (uintptr_t)Y_all) % 16 == 0,
                  "clip_fwd: operand pointers must be 16-byte aligned");
    ClipArgs a = {};
    const ClipPlan pl = clip_plan(Br, Bg, false);
    a.nsplit = pl.nsplit; a.tiles_per_split = pl.tiles_per_split; a.npart = pl.npart;
    PCLIP_REQUIRE(ws_bytes >= clip_fwd_ws(Br, a.npart), "clip_fwd: workspace %zu < %zu bytes", ws_bytes, clip_fwd_ws(Br, a.npart));
    const int npart = a.npart;
    float2* part = (float2*)ws;
    float* diag = (float*)((char*)ws + pclip_align_up((size_t)2 * npart * Br * sizeof(float2), 256));
    a.fin_counter = (unsigned*)((char*)diag + pclip_align_up((size_t)2 * Br * sizeof(float), 256));
    a.fin_part = (float*)((char*)a.fin_counter + 256);
    a.d = d;
    a.logits_mode = logits_out ? logits_mode : 0;
    a.scale = 0.5f / (float)Bg;
    a.p[0] = ClipProb{X_own, Y_all, Br, Bg, row_offset, part, diag, lse_row, a.logits_mode ? logits_out : nullptr, nullptr, nullptr, nullptr};
    a.p[1] = ClipProb{Y_own, X_all, Br, Bg, row_offset, part + (size_t)npart * Br, diag + Br, lse_col, nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    if (pl.tiled) {
        if (int rc = clip_lse_tiled_launch(a, Br, st)) return rc;
    } else {
        hipLaunchKernelGGL(clip_lse_kernel, dim3((Br + 31) / 32, a.nsplit, 2), dim3(256), 0, st, a);
        if (int rc = pclip_check_launch("clip_lse_kernel")) return rc;
    }
    hipLaunchKernelGGL(clip_finalize_kernel, dim3(clip_fin_blocks(Br)), dim3(CLIP_FIN_THREADS), 0, st, a, loss_part);
    return pclip_check_launch("clip_finalize_kernel");
}

// Launches the gradient kernel; slabs land in ws as [2][nsplit][Br][d].
static int clip_grad_launch(const float* X_own, const float* Y_own, const float* X_all, const float* Y_all, const float* lse_row_all,
                            const float* lse_col_all, int Br, int Bg, int d, int row_offset, const float* g, float* slab,
                            int* nsplit_out, hipStream_t st) {
    ClipArgs a = {};
    const ClipPlan pl = clip_plan(Br, Bg, true);
    a.nsplit = pl.nsplit; a.tiles_per_split = pl.tiles_per_split; a.npart = pl.npart;
    a.d = d;
    a.scale = 0.5f / (float)Bg;
    a.g = g;
    a.p[0] = ClipProb{X_own, Y_all, Br, Bg, row_offset, nullptr, nullptr, nullptr, nullptr, lse_row_all + row_offset, lse_col_all, slab};
    a.p[1] = ClipProb{Y_own, X_all, Br, Bg, row_offset, nullptr, nullptr, nullptr, nullptr, lse_col_all + row_offset, lse_row_all,
                      slab + (size_t)a.nsplit * Br * d};
    *nsplit_out = a.nsplit;
    if (pl.tiled) return clip_grad_tiled_launch(a, Br, st);
    static PclipLdsLimit lds_limit;  // 66 KB of dynamic LDS is above the default launch limit
    if (int rc = pclip_ensure_lds(lds_limit, (const void*)clip_grad_kernel, 4 * 32 * 129 * sizeof(float), "clip_grad_kernel")) return rc;
    hipLaunchKernelGGL(clip_grad_kernel, dim3((Br + 31) / 32, a.nsplit, 2), dim3(256), 4 * 32 * 129 * sizeof(float), st, a);
    return pclip_check_launch("clip_grad_kernel");
}

extern "C" int pclip_clip_bwd(const float* X_own, const float* Y_own, const float* X_all, const float* Y_all, const float* lse_row_all,
                              const float* lse_col_all, int Br, int Bg, int d, int row_offset, const float* g, float* dX_own,
                              float* dY_own, void* ws, size_t ws_bytes, void* stream) {
    PCLIP_REQUIRE(X_own && Y_own && X_all && Y_all && lse_row_all && lse_col_all && dX_own && dY_own && ws, "clip_bwd: null pointer");
    if (int rc = clip_check("clip_bwd", Br, Bg, d, row_offset)) return rc;
    PCLIP_REQUIRE(((uintptr_t)X_own | (uintptr_t)Y_own | (uintptr_t)X_all | (uintptr_t)Y_all | (uintptr_t)dX_own | (uintptr_t)dY_own | (uintptr_t)ws) % 16 == 0,
                  "clip_bwd: operand pointers must be 16-byte aligned");
    int ns = clip_plan(Br, Bg, true).nsplit;
    PCLIP_REQUIRE(ws_bytes >= clip_bwd_ws(Br, d, ns), "clip_bwd: workspace %zu < %zu bytes", ws_bytes, clip_bwd_ws(Br, d, ns));
    hipStream_t st = (hipStream_t)stream;
    float* slab = (float*)ws;
    if (int rc = clip_grad_launch(X_own, Y_own, X_all, Y_all, lse_row_all, lse_col_all, Br, Bg, d, row_offset, g, slab, &ns, st)) return rc;
    const size_t n4 = (size_t)Br * d / 4;   // d is a multiple of 4
    const int blocks = (int)((n4 + 255) / 256 > 1024 ? 1024 : (n4 + 255) / 256);
    hipLaunchKernelGGL(slab_sum_kernel, dim3(blocks, 2), dim3(256), 0, st, (const float4*)slab, ns, n4, (float4*)dX_own, (float4*)dY_own, n4);
    return pclip_check_launch("slab_sum_kernel");
}

// ---- fused one-GPU step ----------------------------------------------------------------
// B <= 512: one persistent launch (clip_step.hip).  Larger batches: normalise, scores, gradients,
// normalise backward as separate launches; workspace: x, y [B,d] | lse_row, lse_col [B] | the
// forward partials, later overwritten by the gradient slabs (the partials are dead once
// clip_finalize_kernel has run).
extern "C" size_t pclip_clip_step_workspace_bytes(int B, int d) {
    if (B < 1 || d < 1) return 0;
    if (clip_step_fused_ok(B, d)) return clip_step_fused_ws_bytes(B, d);   // one persistent launch
    return pclip_align_up((size_t)2 * B * d * sizeof(float), 256) + pclip_align_up((size_t)2 * B * sizeof(float), 256) +
           pclip_clip_workspace_bytes(B, B, d);
}

extern "C" int pclip_clip_step(const float* e_pep, const float* e_rec, const float* t, int B, int d, float* loss, float* de_pep,
                               float* de_rec, float* dt, void* ws, size_t ws_bytes, void* stream) {
    PCLIP_REQUIRE(e_pep && e_rec && t && loss && de_pep && de_rec && dt && ws, "clip_step: null pointer");
    if (int rc = clip_check("clip_step", B, B, d, 0)) return rc;
    PCLIP_REQUIRE(ws_bytes >= pclip_clip_step_workspace_bytes(B, d), "clip_step: workspace %zu < %zu bytes", ws_bytes,
                  pclip_clip_step_workspace_bytes(B, d));
    PCLIP_REQUIRE(((uintptr_t)e_pep | (uintptr_t)e_rec | (uintptr_t)de_pep | (uintptr_t)de_rec | (uintptr_t)ws) % 16 == 0,
                  "clip_step: pointers must be 16-byte aligned");
    if (clip_step_fused_ok(B, d)) return clip_step_fused_launch(e_pep, e_rec, t, B, d, loss, de_pep, de_rec, dt, ws, stream);
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)ws;
    float* x = (float*)p;
    float* y = x + (size_t)B * d;
    p += pclip_align_up((size_t)2 * B * d * sizeof(float), 256);
    float* lse_row = (float*)p;
    float* lse_col = lse_row + B;
    p += pclip_align_up((size_t)2 * B * sizeof(float), 256);
    void* cws = p;
    const size_t cws_bytes = pclip_clip_workspace_bytes(B, B, d);
    float* slab = (float*)cws;
    if (int rc = l2norm_scale_fwd_pair_launch(e_pep, e_rec, t, x, y, B, d, dt, stream)) return rc;   // also clears dt
    if (int rc = pclip_clip_fwd(x, y, x, y, B, B, d, 0, lse_row, lse_col, loss, nullptr, 0, cws, cws_bytes, stream)) return rc;
    int ns = 1;
    if (int rc = clip_grad_launch(x, y, x, y, lse_row, lse_col, B, B, d, 0, nullptr, slab, &ns, st)) return rc;
    const size_t n = (size_t)B * d;
    return l2norm_scale_bwd_pair_launch(e_pep, e_rec, t, slab, slab + (size_t)ns * n, ns, n, de_pep, de_rec, dt, B, d, stream);
}

// ---- loss and both gradients from the towers' outputs, one GPU ---------------------------
// B <= 512 on a device that keeps the grid resident: the one persistent launch, without its normalise stages.
// Otherwise the forward and backward launches; workspace: lse_row, lse_col [B] | their scratch.
extern "C" size_t pclip_clip_loss_grad_workspace_bytes(int B, int d) {
    if (B < 1 || d < 1) return 0;
    if (clip_step_fused_ok(B, d)) return clip_step_fused_ws_bytes(B, d);
    return pclip_align_up((size_t)2 * B * sizeof(float), 256) + pclip_clip_workspace_bytes(B, B, d);
}

extern "C" int pclip_clip_loss_grad(const float* x, const float* y, int B, int d, float* loss, float* dx, float* dy, void* ws, size_t ws_bytes,
                                    void* stream) {
    PCLIP_REQUIRE(x && y && loss && dx && dy && ws, "clip_loss_grad: null pointer");
    if (int rc = clip_check("clip_loss_grad", B, B, d, 0)) return rc;
    PCLIP_REQUIRE(ws_bytes >= pclip_clip_loss_grad_workspace_bytes(B, d), "clip_loss_grad: workspace %zu < %zu bytes", ws_bytes,
                  pclip_clip_loss_grad_workspace_bytes(B, d));
    PCLIP_REQUIRE(((uintptr_t)x | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)dy | (uintptr_t)ws) % 16 == 0, "clip_loss_grad: pointers must be 16-byte aligned");
    if (clip_step_fused_ok(B, d)) return clip_step_fused_launch(x, y, nullptr, B, d, loss, dx, dy, nullptr, ws, stream);
    float* lse_row = (float*)ws;
    float* lse_col = lse_row + B;
    void* cws = (char*)ws + pclip_align_up((size_t)2 * B * sizeof(float), 256);
    const size_t cws_bytes = pclip_clip_workspace_bytes(B, B, d);
    if (int rc = pclip_clip_fwd(x, y, x, y, B, B, d, 0, lse_row, lse_col, loss, nullptr, 0, cws, cws_bytes, stream)) return rc;
    return pclip_clip_bwd(x, y, x, y, lse_row, lse_col, B, B, d, 0, nullptr, dx, dy, cws, cws_bytes, stream);
}

// ---- the same step for callers that cannot allocate --------------------------------------
// The workspace of pclip_clip_step lives in a device array of this code object: every GPU the
// library runs on holds its own copy at its own address until the process ends.  Host code never
// uses the array's NAME as a pointer (that is a host-side stand-in, not a device address): the
// address is asked for once per device, with that device current, and kept.
#define CLIP_NOALLOC_SCRATCH_BYTES ((size_t)10 << 20)
static __device__ __align__(256) unsigned char scratch[CLIP_NOALLOC_SCRATCH_BYTES];
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[PCLIP_MAX_DEVICES];   // device address of `scratch` on each device, once resolved

static int clip_scratch_of_current_device(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PCLIP_MAX_DEVICES)
        return pclip_fail(PCLIP_EDEVICE, "clip_step_noalloc: no current device (or more than %d devices)", PCLIP_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return pclip_fail(PCLIP_ELAUNCH, "clip_step_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return PCLIP_OK;
}

extern "C" size_t pclip_clip_step_noalloc_max_bytes(void) { return CLIP_NOALLOC_SCRATCH_BYTES; }

extern "C" int pclip_clip_step_noalloc(const float* e_pep, const float* e_rec, const float* t, int B, int d, float* loss, float* de_pep,
                                       float* de_rec, float* dt, void* stream) {
    PCLIP_REQUIRE(B >= 1 && d >= 1, "clip_step_noalloc: B=%d d=%d", B, d);
    const size_t need = pclip_clip_step_workspace_bytes(B, d);
    PCLIP_REQUIRE(need <= CLIP_NOALLOC_SCRATCH_BYTES,
                  "clip_step_noalloc: B=%d d=%d needs %zu bytes of scratch, the built-in buffer holds %zu (B <= 1024 at d = 128); "
                  "use pclip_clip_step with a caller-provided workspace", B, d, need, CLIP_NOALLOC_SCRATCH_BYTES);
    void* ws = nullptr;
    if (int rc = clip_scratch_of_current_device(&ws)) return rc;
    // every buffer of the step inside the scratch is written before it is read (dt is cleared by the
    // first kernel; the barrier words of the one-launch form live outside the scratch and clear
    // themselves), so nothing has to be zeroed between calls
    return pclip_clip_step(e_pep, e_rec, t, B, d, loss, de_pep, de_rec, dt, ws, CLIP_NOALLOC_SCRATCH_BYTES, stream);
}
