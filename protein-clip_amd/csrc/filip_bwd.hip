// Backward of the FILIP scores (autograd of modules/models.py:332-351), gfx950.
//
// The max over tokens routes every gradient to one (a, b) token pair, so the backward of a
// sequence pair is 2L weighted row additions, not an LxL product.  Each selected pair is
// needed twice: gathered into the gradient of the token that did the choosing, and
// scattered into the gradient of the token that was chosen.  The scatter is turned into a
// gather by a counting sort: for every "own" sequence the partner tokens that chose one of
// its tokens are bucketed by that token (one LDS integer atomic per entry), after which a
// group of 16 lanes accumulates one 128-wide f32 gradient row in registers from 256-B bf16
// partner rows (L2-resident), with no float atomics and no LDS.
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
    int* seg;              // [No][TOp+1] bucket offsets
    uint32_t* list;        // [No][Np*TPp] bucketed entries (partner << 16 | partner token)
    float* dn;             // [No][TOp][128] f32 out
};

// inverse valid-token counts of every sequence
__global__ __launch_bounds__(256) void filip_count_kernel(const uint32_t* __restrict__ mw, int n, int nt, float* __restrict__ ic) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    int c = 0;
    for (int k = 0; k < nt; ++k) c += __popc(mw[(size_t)s * nt + k]);
    ic[s] = c > 0 ? 1.f / (float)c : 0.f;
}

// One workgroup (16 waves) per own sequence: counting sort of the valid partner tokens by the
// own token they chose.  Partners are cut into INV_NCH consecutive chunks and a bucket is laid
// out chunk by chunk, so the gather kernel walks every bucket in (roughly) ascending partner
// order — the same order in which all the other buckets walk — and the partner images it
// touches at any moment stay in the XCD's L2.  Eight u16 entries are read per 16-byte load.
#define INV_NCH 32
__global__ __launch_bounds__(1024) void filip_invert_kernel(FilipBwdArgs p) {
    __shared__ int hist[INV_NCH * PCLIP_FILIP_MAX_T];   // [chunk][own token] counts, then bucket cursors
    __shared__ int tot[PCLIP_FILIP_MAX_T];
    const int tid = threadIdx.x;
    const int o = blockIdx.x;
    const int ntP = p.TPp >> 5;
    const int vpr = p.TPp >> 3;                          // 8-entry vectors per partner row
    const int qpc = (p.Np + INV_NCH - 1) / INV_NCH;      // partners per chunk
    int* seg = p.seg + (size_t)o * (p.TOp + 1);
    for (int k = tid; k < INV_NCH * PCLIP_FILIP_MAX_T; k += 1024) hist[k] = 0;
    __syncthreads();
    const bool own_live = p.icO[o] > 0.f;
    const int nvec = own_live ? p.Np * vpr : 0;
    for (int v = tid; v < nvec; v += 1024) {
        const int q = v / vpr, pt0 = (v - q * vpr) * 8;
        const uint32_t m8 = (p.mwP[(size_t)q * ntP + (pt0 >> 5)] >> (pt0 & 31)) & 0xffu;
        if (m8 == 0) continue;
        const uint4 d = *reinterpret_cast<const uint4*>(p.arg2 + ((size_t)o * p.so + (size_t)q * p.sp) * p.TPp + pt0);
        const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
        int* hc = hist + (q / qpc) * PCLIP_FILIP_MAX_T;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((m8 >> k) & 1u) atomicAdd(&hc[(dw[k >> 1] >> (16 * (k & 1))) & 0xffffu], 1);
    }
    __syncthreads();
    if (tid < PCLIP_FILIP_MAX_T) {
        int s = 0;
        for (int ch = 0; ch < INV_NCH; ++ch) s += hist[ch * PCLIP_FILIP_MAX_T + tid];
        tot[tid] = tid < p.TOp ? s : 0;
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan of the 256 bucket sizes by one wave, 4 per lane
        int v[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = tot[tid * 4 + k];
            s += v[k];
        }
        int incl = s;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
            const int n = __shfl_up(incl, o2);
            if (tid >= o2) incl += n;
        }
        int run = incl - s;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid * 4 + k;
            tot[idx] = run;
            if (idx < p.TOp) seg[idx] = run;
            run += v[k];
        }
        if (tid == 63) seg[p.TOp] = incl;
    }
    __syncthreads();
    if (tid < PCLIP_FILIP_MAX_T) {  // counts -> running cursors, chunk after chunk inside each bucket
        int run = tot[tid];
        for (int ch = 0; ch < INV_NCH; ++ch) {
            const int n = hist[ch * PCLIP_FILIP_MAX_T + tid];
            hist[ch * PCLIP_FILIP_MAX_T + tid] = run;
            run += n;
        }
    }
    __syncthreads();
    uint32_t* list = p.list + (size_t)o * p.Np * p.TPp;
    for (int v = tid; v < nvec; v += 1024) {
        const int q = v / vpr, pt0 = (v - q * vpr) * 8;
        const uint32_t m8 = (p.mwP[(size_t)q * ntP + (pt0 >> 5)] >> (pt0 & 31)) & 0xffu;
        if (m8 == 0) continue;
        const uint4 d = *reinterpret_cast<const uint4*>(p.arg2 + ((size_t)o * p.so + (size_t)q * p.sp) * p.TPp + pt0);
        const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
        int* hc = hist + (q / qpc) * PCLIP_FILIP_MAX_T;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((m8 >> k) & 1u) {
                const int pos = atomicAdd(&hc[(dw[k >> 1] >> (16 * (k & 1))) & 0xffffu], 1);
                list[pos] = ((uint32_t)q << 16) | (uint32_t)(pt0 + k);
            }
    }
}

// 16 lanes own one gradient row (own sequence o, token a): lane l16 keeps columns 8*l16..+7.
// Index/weight loads are spread over the 16 lanes (16 partners or 16 list entries at a time)
// and broadcast inside the group; the 16 row gathers of a batch are all in flight together.
__global__ __launch_bounds__(256) void filip_gather_kernel(FilipBwdArgs p) {
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    const int chunks = p.TOp >> 4;
    const int o = blockIdx.x / chunks;
    const int a = (blockIdx.x - o * chunks) * 16 + grp;
    const bool valid = (p.mwO[(size_t)o * (p.TOp >> 5) + (a >> 5)] >> (a & 31)) & 1u;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (valid) {
        const float inv_t = 1.f / fabsf(p.t[0]);
        const float c1 = p.icO[o] * inv_t;
        // own-side max terms: one partner row per partner sequence
        for (int q0 = 0; q0 < p.Np; q0 += 16) {
            const int q = q0 + l16;
            int b = 0;
            float w = 0.f;
            if (q < p.Np) {
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
                const int qk = min(q0 + k, p.Np - 1);
                rows[k] = *reinterpret_cast<const bf16x8*>(p.hnP + ((size_t)qk * p.TPp + bk) * FD + l16 * 8);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += ws[k] * (float)rows[k][e];
        }
        // partner-side max terms that chose this token: its bucket of the sorted list
        const int* seg = p.seg + (size_t)o * (p.TOp + 1);
        const uint32_t* list = p.list + (size_t)o * p.Np * p.TPp;
        const int end = seg[a + 1];
        for (int pos0 = seg[a]; pos0 < end; pos0 += 16) {
            const int pos = pos0 + l16;
            uint32_t ent = 0;
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
    float4* out = reinterpret_cast<float4*>(p.dn + ((size_t)o * p.TOp + a) * FD + l16 * 8);
    out[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    out[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
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
    size_t listA, listB, segA, segB, icA, icB, total;
};
static BwdWs bwd_ws_layout(int Br, int TA, int Bc, int TB) {
    const size_t TAp = tp32(TA), TBp = tp32(TB);
    BwdWs w;
    size_t off = 0;
    w.listA = off; off += pclip_align_up((size_t)Br * Bc * TBp * 4, 256);   // buckets of B tokens per A sequence
    w.listB = off; off += pclip_align_up((size_t)Bc * Br * TAp * 4, 256);   // buckets of A tokens per B sequence
    w.segA = off;  off += pclip_align_up((size_t)Br * (TAp + 1) * 4, 256);
    w.segB = off;  off += pclip_align_up((size_t)Bc * (TBp + 1) * 4, 256);
    w.icA = off;   off += pclip_align_up((size_t)Br * 4, 256);
    w.icB = off;   off += pclip_align_up((size_t)Bc * 4, 256);
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
    a.icO = icA; a.icP = icB; a.seg = (int*)(w + L.segA); a.list = (uint32_t*)(w + L.listA); a.dn = dnA;
    hipLaunchKernelGGL(filip_invert_kernel, dim3(Br), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(filip_gather_kernel, dim3(Br * (TAp / 16)), dim3(256), 0, st, a);
    if (int rc = pclip_check_launch("filip backward (A side)")) return rc;
    // gradient of the B images: own = B(j), partner = A(i), pair = i*Bc + j
    a.hnP = hnA; a.mwO = maskwB; a.mwP = maskwA; a.No = Bc; a.TOp = TBp; a.Np = Br; a.TPp = TAp;
    a.arg1 = argB; a.arg2 = argA; a.g1 = gB; a.g2 = gA; a.so = 1; a.sp = (size_t)Bc;
    a.icO = icB; a.icP = icA; a.seg = (int*)(w + L.segB); a.list = (uint32_t*)(w + L.listB); a.dn = dnB;
    hipLaunchKernelGGL(filip_invert_kernel, dim3(Bc), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(filip_gather_kernel, dim3(Bc * (TBp / 16)), dim3(256), 0, st, a);
    if (int rc = pclip_check_launch("filip backward (B side)")) return rc;
    const size_t n = (size_t)Br * Bc;
    const int blocks = (int)((n + 255) / 256 > 256 ? 256 : (n + 255) / 256);
    hipLaunchKernelGGL(filip_dt_kernel, dim3(blocks), dim3(256), 0, st, gA, gB, sim_A, sim_B, n, t, dt);
    return pclip_check_launch("filip_dt_kernel");
}
