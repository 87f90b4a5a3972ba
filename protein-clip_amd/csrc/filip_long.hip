// FILIP score kernel for sequences longer than one 256-token LDS image (gfx950).
//
// The reference's two-protein data set keeps chains of up to 2,000 residues
// (modules/data_utils_2protein.py:88-133), so `_filip_similarity_score` (modules/models.py:292-356)
// must work beyond the 256 tokens the fast kernels are built around.  This kernel is the
// general one: any Tp up to PCLIP_FILIP_MAX_T_LONG, with or without unit-norm operands, exact
// f32 maxima with the lowest index winning ties.  It is not tuned — batches of such chains are
// small and the trunk dominates their step — but it is still MFMA work with no [TA,TB] tensor
// in memory: one workgroup per sequence pair (i, j); wave w owns the A tiles w, w+4, ...; for
// each of them it walks every valid B tile (fragments straight from L2), keeps the running row
// maxima in registers and the running column maxima of ITS A tiles in an LDS array of its own;
// the four arrays are merged at the end.
#include "pclip_common.h"

#define FD 128

struct FilipLongArgs {
    const uint16_t* hnA; const uint32_t* mwA; int Br, TAp;
    const uint16_t* hnB; const uint32_t* mwB; int Bc, TBp;
    const float* t;
    float* simA; float* simB;
    uint16_t* argA; uint16_t* argB;
    int ldc, col0;   // the outputs are [Br, ldc(, T)] arrays of which this launch fills the columns col0 .. col0 + Bc - 1
};

__global__ __launch_bounds__(256) void filip_sim_long_kernel(FilipLongArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* colst = reinterpret_cast<float2*>(smem);                            // [4][TBp] (best value, arg bits)
    float* red = reinterpret_cast<float*>(smem + (size_t)4 * p.TBp * sizeof(float2));  // [8]
    int* cnt = reinterpret_cast<int*>(red + 8);                                 // [2] valid tokens of A(i), B(j)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int i = blockIdx.x / p.Bc, j = blockIdx.x - i * p.Bc;
    const int ntA = p.TAp >> 5, ntB = p.TBp >> 5;
    const uint32_t* mwA = p.mwA + (size_t)i * ntA;
    const uint32_t* mwB = p.mwB + (size_t)j * ntB;

    for (int k = tid; k < 4 * p.TBp; k += 256) colst[k] = make_float2(-FLT_MAX, __int_as_float(0));
    if (w < 2) {  // valid-token counts of the two sequences
        const uint32_t* mw = w == 0 ? mwA : mwB;
        const int nt = w == 0 ? ntA : ntB;
        int n = 0;
        for (int k = lane; k < nt; k += 64) n += __popc(mw[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
        if (lane == 0) cnt[w] = n;
    }
    __syncthreads();

    float2* mycol = colst + (size_t)w * p.TBp;
    float rs = 0.f;
    for (int at = w; at < ntA; at += 4) {
        const uint32_t wA = mwA[at];
        if (wA == 0u) continue;   // a tile of padding: its arg-max entries are unspecified
        bf16x8 afr[8];
        {
            const uint16_t* src = p.hnA + (((size_t)i * p.TAp + at * 32 + c) * FD + 8 * h);
#pragma unroll
            for (int s = 0; s < 8; ++s) afr[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
        }
        float M[16];
        int MI[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { M[r] = -FLT_MAX; MI[r] = 0; }
        for (int bt = 0; bt < ntB; ++bt) {
            const uint32_t wB = mwB[bt];
            if (wB == 0u) continue;
            bf16x8 bfr[8];
            {
                const uint16_t* src = p.hnB + (((size_t)j * p.TBp + bt * 32 + c) * FD + 8 * h);
#pragma unroll
                for (int s = 0; s < 8; ++s) bfr[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
            }
            // padded pairs start (and stay) at -FLT_MAX: finfo.min of models.py:336-338
            f32x16 acc;
            const float colbias = ((wB >> c) & 1u) ? 0.f : -FLT_MAX;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = ((wA >> acc_row(r, h)) & 1u) ? colbias : -FLT_MAX;
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[s], bfr[s], acc, 0, 0, 0);
            float cbest = -FLT_MAX;
            int carg = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool up = acc[r] > M[r];
                M[r] = up ? acc[r] : M[r];
                MI[r] = up ? bt : MI[r];
                const bool upc = acc[r] > cbest;
                cbest = upc ? acc[r] : cbest;
                carg = upc ? (at * 32 + acc_row(r, h)) : carg;
            }
            const float ob = __shfl_xor(cbest, 32);
            const int oa = __shfl_xor(carg, 32);
            if (ob > cbest || (ob == cbest && oa < carg)) { cbest = ob; carg = oa; }
            if (h == 0) {  // this wave's A tiles come in ascending order: an equal value keeps the earlier token
                float2* slot = mycol + bt * 32 + c;
                if (cbest > slot->x) *slot = make_float2(cbest, __int_as_float(carg));
            }
        }
        // rows of this A tile: reduce each register across the 32 lanes of its half (value + arg)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = M[r];
            int b = MI[r] * 32 + c;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const float ov = __shfl_xor(v, o);
                const int ob = __shfl_xor(b, o);
                if (ov > v || (ov == v && ob < b)) { v = ov; b = ob; }
            }
            const int arow = acc_row(r, h);
            const bool valid = (wA >> arow) & 1u;
            rs += (valid && c == 0) ? v : 0.f;
            if (p.argA && c == r) p.argA[((size_t)i * p.ldc + p.col0 + j) * p.TAp + at * 32 + arow] = valid ? (uint16_t)b : (uint16_t)0;
        }
    }
    rs = wave_sum(rs);
    if (lane == 0) red[w] = rs;
    __syncthreads();

    // columns: merge the four waves' maxima (lowest token index on ties), masked sum
    float cv = 0.f;
    for (int b = tid; b < p.TBp; b += 256) {
        const bool valid = (mwB[b >> 5] >> (b & 31)) & 1u;
        float best = -FLT_MAX;
        int arg = 0;
        if (valid) {
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const float2 q = colst[(size_t)ww * p.TBp + b];
                const int qa = __float_as_int(q.y);
                if (q.x > best || (q.x == best && qa < arg)) { best = q.x; arg = qa; }
            }
            cv += best;
            if (p.argB) p.argB[((size_t)i * p.ldc + p.col0 + j) * p.TBp + b] = (uint16_t)arg;
        }
    }
    cv = wave_sum(cv);
    if (lane == 0) red[4 + w] = cv;
    __syncthreads();
    if (tid == 0) {
        const float inv_t = 1.f / fabsf(p.t[0]);
        const float sa = red[0] + red[1] + red[2] + red[3];
        const float sb = red[4] + red[5] + red[6] + red[7];
        p.simA[(size_t)i * p.ldc + p.col0 + j] = sa / fmaxf((float)cnt[0], 1e-6f) * inv_t;
        p.simB[(size_t)i * p.ldc + p.col0 + j] = sb / fmaxf((float)cnt[1], 1e-6f) * inv_t;
    }
}

int filip_sim_long_launch(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TAp, const uint16_t* hnB, const uint32_t* maskwB, int Bc,
                          int TBp, const float* t, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB, int col_offset, int ldc,
                          void* stream) {
    FilipLongArgs a;
    a.hnA = hnA; a.mwA = maskwA; a.Br = Br; a.TAp = TAp;
    a.hnB = hnB; a.mwB = maskwB; a.Bc = Bc; a.TBp = TBp;
    a.t = t; a.simA = sim_A; a.simB = sim_B; a.argA = argA; a.argB = argB;
    a.ldc = ldc; a.col0 = col_offset;
    const size_t lds = (size_t)4 * TBp * sizeof(float2) + 8 * sizeof(float) + 2 * sizeof(int);
    static PclipLdsLimit lds_limit;
    if (int rc = pclip_ensure_lds(lds_limit, (const void*)filip_sim_long_kernel, lds, "filip_sim_long_kernel")) return rc;
    hipLaunchKernelGGL(filip_sim_long_kernel, dim3((unsigned)((size_t)Br * Bc)), dim3(256), lds, (hipStream_t)stream, a);
    return pclip_check_launch("filip_sim_long_kernel");
}

// ---- long sequences on the streaming kernel ------------------------------------------------------------------------
// Unit-norm operands: every sequence is taken as segments of 256 tokens (its image [Tp][128] and its mask words
// already are rows of such segments, the last one possibly short), the streaming kernel of
// filip_stream.hip scores every (A segment, B segment) pair at its usual rate and leaves, per segment pair, the
// maximum and its partner for every token, and a second kernel joins the segments of a sequence pair: for an A token
// the best of its maxima over the B segments (the later segment on equal values, as the later tile wins inside the
// streaming kernel), masked means as in modules/models.py:345-351.  The workspace holds the per-segment-pair tables;
// the columns are worked off in blocks so that it stays within a fixed budget whatever the batch.
#define LONG_SEG 256
#define LONG_WS_BUDGET ((size_t)256 << 20)

struct FilipJoinArgs {
    const float* rowval; const uint16_t* rowarg; const float* colval; const uint16_t* colarg;   // [Br * segA][ldv][256]
    const uint32_t* mwA; const uint32_t* mwB;                                                   // real mask words [Br][ntA], [Bc][ntB]
    const float* t;
    float* simA; float* simB; uint16_t* argA; uint16_t* argB;                                   // real outputs [Br][ldc](, Tp)
    int Br, TAp, segA, Bc, TBp, segB;   // Bc: sequences of this column block
    int ldv;                            // segment columns of the tables (= block sequences * segB)
    int col0, ldc, jb0;                 // the block's first output column, output row pitch, first B sequence of the block (mask rows)
};

// The best of a token's maxima over the segments of the other side: all loads of a token in flight together
// (at most 16 segments: 4096 tokens), the later segment on equal values.
#define LONG_MAX_SEG 16
__device__ __forceinline__ float join_token(const float* __restrict__ val, const uint16_t* __restrict__ arg, size_t stride, int nseg, int* out_arg) {
    float v[LONG_MAX_SEG];
    uint16_t g[LONG_MAX_SEG];
#pragma unroll
    for (int s = 0; s < LONG_MAX_SEG; ++s) {
        v[s] = s < nseg ? val[(size_t)s * stride] : -FLT_MAX;
        g[s] = s < nseg ? arg[(size_t)s * stride] : (uint16_t)0;
    }
    float best = -FLT_MAX;
    int a = 0;
#pragma unroll
    for (int s = 0; s < LONG_MAX_SEG; ++s)
        if (v[s] >= best && v[s] > -FLT_MAX) { best = v[s]; a = s * LONG_SEG + (int)g[s]; }
    *out_arg = a;
    return best;
}

// One workgroup of 1024 threads per sequence pair (i, j): a thread per token of either side.
__global__ __launch_bounds__(1024) void filip_long_join_kernel(FilipJoinArgs p) {
    __shared__ float red[32];
    __shared__ int cnt[32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = blockIdx.x / p.Bc, j = blockIdx.x - i * p.Bc;
    const int ntA = p.TAp >> 5, ntB = p.TBp >> 5;
    const uint32_t* mwA = p.mwA + (size_t)i * ntA;
    const uint32_t* mwB = p.mwB + (size_t)(p.jb0 + j) * ntB;
    const size_t out_pair = (size_t)i * p.ldc + p.col0 + j;
    float sa = 0.f, sb = 0.f;
    int na = 0, nb = 0;
    // (the tables are read whether or not the token is valid — entries of padding tokens are inside the tables, just
    // meaningless — so that the mask word and the table entries are one round trip, not two)
    for (int a = tid; a < p.TAp; a += 1024) {   // rows: over the B segments (consecutive in the tables)
        const uint32_t wd = mwA[a >> 5];
        const int s = a / LONG_SEG, al = a - s * LONG_SEG;
        const size_t o = ((size_t)(i * p.segA + s) * p.ldv + j * p.segB) * LONG_SEG + al;
        int arg;
        const float best = join_token(p.rowval + o, p.rowarg + o, LONG_SEG, p.segB, &arg);
        if (!((wd >> (a & 31)) & 1u)) continue;
        sa += best;
        ++na;
        if (p.argA) p.argA[out_pair * p.TAp + a] = (uint16_t)arg;
    }
    for (int b = tid; b < p.TBp; b += 1024) {   // columns: over the A segments (a table row apart)
        const uint32_t wd = mwB[b >> 5];
        const int s = b / LONG_SEG, bl = b - s * LONG_SEG;
        const size_t o = ((size_t)(i * p.segA) * p.ldv + j * p.segB + s) * LONG_SEG + bl;
        int arg;
        const float best = join_token(p.colval + o, p.colarg + o, (size_t)p.ldv * LONG_SEG, p.segA, &arg);
        if (!((wd >> (b & 31)) & 1u)) continue;
        sb += best;
        ++nb;
        if (p.argB) p.argB[out_pair * p.TBp + b] = (uint16_t)arg;
    }
    sa = wave_sum(sa);
    sb = wave_sum(sb);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { na += __shfl_xor(na, o); nb += __shfl_xor(nb, o); }
    if (lane == 0) { red[w] = sa; red[16 + w] = sb; cnt[w] = na; cnt[16 + w] = nb; }
    __syncthreads();
    if (tid == 0) {
        const float inv_t = 1.f / fabsf(p.t[0]);
        float ta = 0.f, tb = 0.f;
        int ca = 0, cb = 0;
        for (int k = 0; k < 16; ++k) { ta += red[k]; tb += red[16 + k]; ca += cnt[k]; cb += cnt[16 + k]; }
        p.simA[out_pair] = ta / fmaxf((float)ca, 1e-6f) * inv_t;
        p.simB[out_pair] = tb / fmaxf((float)cb, 1e-6f) * inv_t;
    }
}

struct FilipLongPlan { int segA, segB, blk; size_t rowval, colval, rowarg, colarg, sims, total; };

static FilipLongPlan filip_long_plan(int Br, int TAp, int Bc, int TBp) {
    FilipLongPlan P;
    P.segA = (TAp + LONG_SEG - 1) / LONG_SEG;
    P.segB = (TBp + LONG_SEG - 1) / LONG_SEG;
    const size_t brv = (size_t)Br * P.segA;
    // per B sequence of a column block: two tables of (f32 + u16) per token of every (A segment, B segment) pair, and two dummy scores
    const size_t per_seq = brv * P.segB * (LONG_SEG * 12 + 8);
    size_t blk = LONG_WS_BUDGET / per_seq;
    if (blk < 1) blk = 1;
    if (blk > (size_t)Bc) blk = (size_t)Bc;
    P.blk = (int)blk;
    const size_t pairs = brv * blk * P.segB;
    size_t off = 0;
    P.rowval = off; off += pclip_align_up(pairs * LONG_SEG * 4, 256);
    P.colval = off; off += pclip_align_up(pairs * LONG_SEG * 4, 256);
    P.rowarg = off; off += pclip_align_up(pairs * LONG_SEG * 2, 256);
    P.colarg = off; off += pclip_align_up(pairs * LONG_SEG * 2, 256);
    P.sims = off; off += pclip_align_up(pairs * 2 * 4, 256);
    P.total = off;
    return P;
}

size_t filip_sim_long_stream_ws_bytes(int Br, int TAp, int Bc, int TBp) { return filip_long_plan(Br, TAp, Bc, TBp).total; }

int filip_sim_long_stream_launch(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TAp, const uint16_t* hnB, const uint32_t* maskwB, int Bc,
                                 int TBp, const float* t, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB, int col_offset, int ldc,
                                 void* ws, void* stream) {
    const FilipLongPlan P = filip_long_plan(Br, TAp, Bc, TBp);
    char* w = (char*)ws;
    hipStream_t st = (hipStream_t)stream;
    const int brv = Br * P.segA;
    for (int jb0 = 0; jb0 < Bc; jb0 += P.blk) {
        const int nb = Bc - jb0 < P.blk ? Bc - jb0 : P.blk;
        const int ldv = nb * P.segB;
        float* sims = (float*)(w + P.sims);
        // the block's B segments: images from sequence jb0 on, mask rows from segment jb0 * segB on
        if (int rc = filip_sim_stream_launch_seg(hnA, maskwA, brv, LONG_SEG, P.segA, TAp, hnB + (size_t)jb0 * TBp * FD, maskwB + (size_t)jb0 * (TBp >> 5),
                                                 ldv, LONG_SEG, P.segB, TBp >> 5, t, sims, sims + (size_t)brv * ldv, (uint16_t*)(w + P.rowarg),
                                                 (uint16_t*)(w + P.colarg), (float*)(w + P.rowval), (float*)(w + P.colval), 0, ldv, 0, stream))
            return rc;
        FilipJoinArgs a;
        a.rowval = (const float*)(w + P.rowval); a.rowarg = (const uint16_t*)(w + P.rowarg);
        a.colval = (const float*)(w + P.colval); a.colarg = (const uint16_t*)(w + P.colarg);
        a.mwA = maskwA; a.mwB = maskwB; a.t = t;
        a.simA = sim_A; a.simB = sim_B; a.argA = argA; a.argB = argB;
        a.Br = Br; a.TAp = TAp; a.segA = P.segA; a.Bc = nb; a.TBp = TBp; a.segB = P.segB;
        a.ldv = ldv; a.col0 = col_offset + jb0; a.ldc = ldc; a.jb0 = jb0;
        hipLaunchKernelGGL(filip_long_join_kernel, dim3((unsigned)((size_t)Br * nb)), dim3(1024), 0, st, a);
        if (int rc = pclip_check_launch("filip_long_join_kernel")) return rc;
    }
    return PCLIP_OK;
}
