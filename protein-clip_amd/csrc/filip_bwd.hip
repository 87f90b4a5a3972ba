// Backward of the FILIP scores (autograd of modules/models.py:332-351), gfx950.
//
// The max over tokens routes every gradient to one (a, b) token pair, so the backward of a
// sequence pair is 2L weighted row additions, not an LxL product.  Each selected pair is
// needed twice: gathered into the gradient of the token that did the choosing, and
// scattered into the gradient of the token that was chosen.  The scatter is turned into a
// gather by a counting sort: for every "own" sequence the partner tokens that chose one of
// its tokens are bucketed by that token (one LDS integer atomic per entry), after which a
// group of 16 lanes accumulates one 128-wide f32 gradient row in registers from 256-B bf16
// partner rows, with no float atomics and no LDS.
//
// The partner images of a whole batch (tens of MB) do not fit the 4-MiB L2 of an XCD, and
// row gathers served from beyond L2 run at half the rate.  So the partners are cut into
// BWD_NWIN windows; workgroup b works on window b % 8 — blocks b and b + 8 are dispatched to
// the same XCD, so each XCD's L2 only ever sees one window's images — and writes a PARTIAL
// gradient row; a last pass sums the partials of the eight windows.
#include "pclip_common.h"

#define FD 128

struct FilipBwdArgs {
    const uint16_t* hnP;   // partner images [Np][TPp][128] bf16
    const uint32_t* mwO;   // own mask words [No][TOp/32]
    const uint32_t* mwP;   // partner mask words [Np][TPp/32]
    int No, TOp, Np, TPp;
    const uint16_t* arg1;  // [pair][TOp]  partner token chosen by each own token
    const uint16_t* arg2;  // [pair][TPp]  own token chosen by each partner token
    const float* g1;       // [pair] d loss / d score averaged over OWN tokens
    const float* g2;       // [pair] d loss / d score averaged over PARTNER tokens
    size_t so, sp;         // pair index = own*so + partner*sp
    const float* t;
    const float* icO;      // [No] 1/max(valid own tokens, 1e-6), 0 where the sequence is empty
    const float* icP;      // [Np] same for partners
    int* seg;              // [No][BWD_NWIN*TOp+1] bucket offsets, index = window*TOp + own token
    uint32_t* list;        // [No][Np*TPp] bucketed entries (partner << 16 | partner token)
    float* dn;             // [nwin][No][TOp][128] f32 out: one partial gradient per partner window
    int nwin, qpw;         // partner windows in use (1 or BWD_NWIN), partners per window
};

// inverse valid-token counts of every sequence
__global__ __launch_bounds__(256) void filip_count_kernel(const uint32_t* __restrict__ mw, int n, int nt, float* __restrict__ ic) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    int c = 0;
    for (int k = 0; k < nt; ++k) c += __popc(mw[(size_t)s * nt + k]);
    ic[s] = c > 0 ? 1.f / (float)c : 0.f;
}

// One workgroup (16 waves) per own sequence: counting sort of the valid partner tokens by
// (partner window, own token they chose).  Eight u16 entries are read per 16-byte load.
#define BWD_NWIN 8
static_assert(BWD_NWIN * PCLIP_FILIP_MAX_T == 2048, "the scan below gives two buckets to each of 1024 threads");
__global__ __launch_bounds__(1024) void filip_invert_kernel(FilipBwdArgs p) {
    __shared__ int hist[BWD_NWIN * PCLIP_FILIP_MAX_T];   // [window][own token] counts, then bucket cursors
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int o = blockIdx.x;
    const int ntP = p.TPp >> 5;
    const int vpr = p.TPp >> 3;                          // 8-entry vectors per partner row
    int* seg = p.seg + (size_t)o * (BWD_NWIN * p.TOp + 1);
    hist[2 * tid] = 0;
    hist[2 * tid + 1] = 0;
    __syncthreads();
    const bool own_live = p.icO[o] > 0.f;
    const int nvec = own_live ? p.Np * vpr : 0;
    for (int v = tid; v < nvec; v += 1024) {
        const int q = v / vpr, pt0 = (v - q * vpr) * 8;
        const uint32_t m8 = (p.mwP[(size_t)q * ntP + (pt0 >> 5)] >> (pt0 & 31)) & 0xffu;
        if (m8 == 0) continue;
        const uint4 d = *reinterpret_cast<const uint4*>(p.arg2 + ((size_t)o * p.so + (size_t)q * p.sp) * p.TPp + pt0);
        const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
        int* hc = hist + (q / p.qpw) * PCLIP_FILIP_MAX_T;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((m8 >> k) & 1u) atomicAdd(&hc[(dw[k >> 1] >> (16 * (k & 1))) & 0xffffu], 1);
    }
    __syncthreads();
    {   // exclusive scan of the 2048 counts in (window, token) order, two per thread
        const int v0 = hist[2 * tid], v1 = hist[2 * tid + 1];
        const int s = v0 + v1;
        int incl = s;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
            const int n = __shfl_up(incl, o2);
            if (lane >= o2) incl += n;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < w; ++k) base += wsum[k];
        const int e0 = base + incl - s, e1 = e0 + v0;
        hist[2 * tid] = e0;
        hist[2 * tid + 1] = e1;
        // counts of tokens >= TOp are zero, so their offsets equal the next stored one
        const int x = (2 * tid) / PCLIP_FILIP_MAX_T, a0 = (2 * tid) % PCLIP_FILIP_MAX_T;
        if (a0 < p.TOp) seg[x * p.TOp + a0] = e0;
        if (a0 + 1 < p.TOp) seg[x * p.TOp + a0 + 1] = e1;
        if (tid == 1023) seg[BWD_NWIN * p.TOp] = base + incl;
    }
    __syncthreads();
    uint32_t* list = p.list + (size_t)o * p.Np * p.TPp;
    for (int v = tid; v < nvec; v += 1024) {
        const int q = v / vpr, pt0 = (v - q * vpr) * 8;
        const uint32_t m8 = (p.mwP[(size_t)q * ntP + (pt0 >> 5)] >> (pt0 & 31)) & 0xffu;
        if (m8 == 0) continue;
        const uint4 d = *reinterpret_cast<const uint4*>(p.arg2 + ((size_t)o * p.so + (size_t)q * p.sp) * p.TPp + pt0);
        const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
        int* hc = hist + (q / p.qpw) * PCLIP_FILIP_MAX_T;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((m8 >> k) & 1u) {
                const int pos = atomicAdd(&hc[(dw[k >> 1] >> (16 * (k & 1))) & 0xffffu], 1);
                list[pos] = ((uint32_t)q << 16) | (uint32_t)(pt0 + k);
            }
    }
}

// 16 lanes own one gradient row (own sequence o, token a) for ONE partner window: lane l16
// keeps columns 8*l16..+7.  Index/weight loads are spread over the 16 lanes (16 partners or 16
// list entries at a time) and broadcast inside the group; the 16 row gathers of a batch are all
// in flight together.
__global__ __launch_bounds__(256) void filip_gather_kernel(FilipBwdArgs p) {
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    const int chunks = p.TOp >> 4;
    const int x = blockIdx.x % p.nwin, blk = blockIdx.x / p.nwin;
    const int o = blk / chunks;
    const int a = (blk - o * chunks) * 16 + grp;
    const int q_lo = x * p.qpw, q_hi = min(p.Np, q_lo + p.qpw);
    const bool valid = (p.mwO[(size_t)o * (p.TOp >> 5) + (a >> 5)] >> (a & 31)) & 1u;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (valid && q_lo < q_hi) {
        const float inv_t = 1.f / fabsf(p.t[0]);
        const float c1 = p.icO[o] * inv_t;
        // own-side max terms: one partner row per partner sequence of the window
        for (int q0 = q_lo; q0 < q_hi; q0 += 16) {
            const int q = q0 + l16;
            int b = 0;
            float w = 0.f;
            if (q < q_hi) {
                const size_t pair = (size_t)o * p.so + (size_t)q * p.sp;
                b = p.arg1[pair * p.TOp + a];
                w = p.icP[q] > 0.f ? p.g1[pair] * c1 : 0.f;
            }
            bf16x8 rows[16];
            float ws[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int bk = __shfl(b, k, 16);
                ws[k] = __shfl(w, k, 16);
                const int qk = min(q0 + k, q_hi - 1);
                rows[k] = *reinterpret_cast<const bf16x8*>(p.hnP + ((size_t)qk * p.TPp + bk) * FD + l16 * 8);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += ws[k] * (float)rows[k][e];
        }
        // partner-side max terms of this window that chose this token: its bucket of the sorted list
        const int* seg = p.seg + (size_t)o * (BWD_NWIN * p.TOp + 1) + x * p.TOp + a;
        const uint32_t* list = p.list + (size_t)o * p.Np * p.TPp;
        const int end = seg[1];
        for (int pos0 = seg[0]; pos0 < end; pos0 += 16) {
            const int pos = pos0 + l16;
            uint32_t ent = (uint32_t)q_lo << 16;   // filler: weight 0, a row inside the window
            float w = 0.f;
            if (pos < end) {
                ent = list[pos];
                const int q = ent >> 16;
                w = p.g2[(size_t)o * p.so + (size_t)q * p.sp] * p.icP[q] * inv_t;
            }
            bf16x8 rows[16];
            float ws[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t ek = __shfl(ent, k, 16);
                ws[k] = __shfl(w, k, 16);
                rows[k] = *reinterpret_cast<const bf16x8*>(p.hnP + ((size_t)(ek >> 16) * p.TPp + (ek & 0xffffu)) * FD + l16 * 8);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += ws[k] * (float)rows[k][e];
        }
    }
    float4* out = reinterpret_cast<float4*>(p.dn + (((size_t)x * p.No + o) * p.TOp + a) * FD + l16 * 8);
    out[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    out[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// out = sum of the nwin partial gradients (n4 float4 each)
__global__ __launch_bounds__(256) void filip_window_sum_kernel(const float4* __restrict__ part, size_t n4, int nwin, float4* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n4) return;
    float4 s = part[k];
    for (int x = 1; x < nwin; ++x) {
        const float4 v = part[(size_t)x * n4 + k];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[k] = s;
}

// dt += -(1/t) * sum(gA*simA + gB*simB); pairs whose partner is empty hold infinities in the
// reference too and are left out.
__global__ __launch_bounds__(256) void filip_dt_kernel(const float* __restrict__ gA, const float* __restrict__ gB, const float* __restrict__ sA,
                                                       const float* __restrict__ sB, size_t n, const float* __restrict__ t,
                                                       float* __restrict__ dt) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) acc += gA[k] * sA[k] + gB[k] * sB[k];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dt, -(red[0] + red[1] + red[2] + red[3]) / t[0]);
}

// ---- host side -------------------------------------------------------------------------
static inline int tp32(int T) { return (T + 31) / 32 * 32; }

struct BwdWs {
    size_t listA, listB, segA, segB, icA, icB, partA, partB, total;
    int nwinA, nwinB;   // partner windows of the A-side pass (partners = B) and of the B-side pass
};
// One window when the partner images fit an XCD's L2 with room to spare, else BWD_NWIN.
static int bwd_windows(int Np, int TPp) { return (size_t)Np * TPp * FD * 2 > ((size_t)3 << 19) && Np >= BWD_NWIN ? BWD_NWIN : 1; }
static BwdWs bwd_ws_layout(int Br, int TA, int Bc, int TB) {
    const size_t TAp = tp32(TA), TBp = tp32(TB);
    BwdWs w;
    size_t off = 0;
    w.listA = off; off += pclip_align_up((size_t)Br * Bc * TBp * 4, 256);   // buckets of B tokens per A sequence
    w.listB = off; off += pclip_align_up((size_t)Bc * Br * TAp * 4, 256);   // buckets of A tokens per B sequence
    w.segA = off;  off += pclip_align_up((size_t)Br * (BWD_NWIN * TAp + 1) * 4, 256);
    w.segB = off;  off += pclip_align_up((size_t)Bc * (BWD_NWIN * TBp + 1) * 4, 256);
    w.icA = off;   off += pclip_align_up((size_t)Br * 4, 256);
    w.icB = off;   off += pclip_align_up((size_t)Bc * 4, 256);
    w.nwinA = bwd_windows(Bc, (int)TBp);
    w.nwinB = bwd_windows(Br, (int)TAp);
    w.partA = off; off += w.nwinA > 1 ? pclip_align_up((size_t)w.nwinA * Br * TAp * FD * 4, 256) : 0;   // window partials of dnA
    w.partB = off; off += w.nwinB > 1 ? pclip_align_up((size_t)w.nwinB * Bc * TBp * FD * 4, 256) : 0;
    w.total = off;
    return w;
}

extern "C" size_t pclip_filip_bwd_workspace_bytes(int Br, int TA, int Bc, int TB) {
    if (Br < 1 || Bc < 1 || TA < 1 || TB < 1) return 0;
    return bwd_ws_layout(Br, TA, Bc, TB).total;
}

extern "C" int pclip_filip_sim_bwd(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA, const uint16_t* hnB, const uint32_t* maskwB,
                                   int Bc, int TB, const float* t, const float* sim_A, const float* sim_B, const uint16_t* argA,
                                   const uint16_t* argB, const float* gA, const float* gB, float* dnA, float* dnB, float* dt, void* ws,
                                   size_t ws_bytes, void* stream) {
    PCLIP_REQUIRE(hnA && maskwA && hnB && maskwB && t && sim_A && sim_B && argA && argB && gA && gB && dnA && dnB && dt && ws,
                  "filip_sim_bwd: null pointer");
    PCLIP_REQUIRE(Br >= 1 && Bc >= 1 && Br <= 65535 && Bc <= 65535, "filip_sim_bwd: Br=%d Bc=%d (need 1..65535)", Br, Bc);
    PCLIP_REQUIRE(TA >= 1 && TA <= PCLIP_FILIP_MAX_T && TB >= 1 && TB <= PCLIP_FILIP_MAX_T, "filip_sim_bwd: TA=%d TB=%d (need 1..%d)", TA, TB,
                  PCLIP_FILIP_MAX_T);
    const BwdWs L = bwd_ws_layout(Br, TA, Bc, TB);
    PCLIP_REQUIRE(ws_bytes >= L.total, "filip_sim_bwd: workspace %zu < %zu bytes", ws_bytes, L.total);
    const int TAp = tp32(TA), TBp = tp32(TB);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* icA = (float*)(w + L.icA);
    float* icB = (float*)(w + L.icB);
    hipLaunchKernelGGL(filip_count_kernel, dim3((Br + 255) / 256), dim3(256), 0, st, maskwA, Br, TAp / 32, icA);
    hipLaunchKernelGGL(filip_count_kernel, dim3((Bc + 255) / 256), dim3(256), 0, st, maskwB, Bc, TBp / 32, icB);
    FilipBwdArgs a;
    // gradient of the A images: own = A(i), partner = B(j), pair = i*Bc + j
    a.hnP = hnB; a.mwO = maskwA; a.mwP = maskwB; a.No = Br; a.TOp = TAp; a.Np = Bc; a.TPp = TBp;
    a.arg1 = argA; a.arg2 = argB; a.g1 = gA; a.g2 = gB; a.so = (size_t)Bc; a.sp = 1; a.t = t;
    a.icO = icA; a.icP = icB; a.seg = (int*)(w + L.segA); a.list = (uint32_t*)(w + L.listA);
    a.nwin = L.nwinA; a.qpw = a.nwin > 1 ? (Bc + BWD_NWIN - 1) / BWD_NWIN : Bc;
    a.dn = a.nwin > 1 ? (float*)(w + L.partA) : dnA;
    hipLaunchKernelGGL(filip_invert_kernel, dim3(Br), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(filip_gather_kernel, dim3(Br * (TAp / 16) * a.nwin), dim3(256), 0, st, a);
    if (a.nwin > 1) {
        const size_t n4 = (size_t)Br * TAp * FD / 4;
        hipLaunchKernelGGL(filip_window_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float4*)a.dn, n4, a.nwin, (float4*)dnA);
    }
    if (int rc = pclip_check_launch("filip backward (A side)")) return rc;
    // gradient of the B images: own = B(j), partner = A(i), pair = i*Bc + j
    a.hnP = hnA; a.mwO = maskwB; a.mwP = maskwA; a.No = Bc; a.TOp = TBp; a.Np = Br; a.TPp = TAp;
    a.arg1 = argB; a.arg2 = argA; a.g1 = gB; a.g2 = gA; a.so = 1; a.sp = (size_t)Bc;
    a.icO = icB; a.icP = icA; a.seg = (int*)(w + L.segB); a.list = (uint32_t*)(w + L.listB);
    a.nwin = L.nwinB; a.qpw = a.nwin > 1 ? (Br + BWD_NWIN - 1) / BWD_NWIN : Br;
    a.dn = a.nwin > 1 ? (float*)(w + L.partB) : dnB;
    hipLaunchKernelGGL(filip_invert_kernel, dim3(Bc), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(filip_gather_kernel, dim3(Bc * (TBp / 16) * a.nwin), dim3(256), 0, st, a);
    if (a.nwin > 1) {
        const size_t n4 = (size_t)Bc * TBp * FD / 4;
        hipLaunchKernelGGL(filip_window_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float4*)a.dn, n4, a.nwin, (float4*)dnB);
    }
    if (int rc = pclip_check_launch("filip backward (B side)")) return rc;
    const size_t n = (size_t)Br * Bc;
    const int blocks = (int)((n + 255) / 256 > 256 ? 256 : (n + 255) / 256);
    hipLaunchKernelGGL(filip_dt_kernel, dim3(blocks), dim3(256), 0, st, gA, gB, sim_A, sim_B, n, t, dt);
    return pclip_check_launch("filip_dt_kernel");
}
