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
