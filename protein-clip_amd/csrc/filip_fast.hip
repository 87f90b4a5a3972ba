// FILIP score kernel for UNIT-NORM operands (the ExtendedFILIP.forward path: tokens are
// L2-normalised before scoring, modules/models.py:238-241), gfx950.
//
// Because |a.b| <= 1, the MFMA accumulator is started at 2.0 so every valid similarity is a
// positive float in [1,3]; positive floats order like their bit patterns, so a running
// maximum AND its arg-max are kept in ONE signed-integer max per element by overwriting the
// low mantissa bits with the index (3 bits of b-tile for the row maximum, 4 bits of
// accumulator register for the column maximum).  Masked pairs are set to -FLT_MAX (negative
// as an integer, never wins).  The dropped mantissa bits are replaced by their midpoint, so a
// score is within 4 (row side) / 8 (column side) f32 ulps of [1,4) — under 2e-6 — of the
// exact maximum, unbiased.
//
// Workgroup = 8 waves = two A sequences; their valid 32-token tiles are dealt round-robin to
// the waves (balanced over the 4 SIMDs) and stay in registers as MFMA A operands.  B
// sequences stream through one XOR-swizzled 64-KB LDS image; the next image is prefetched
// into registers while the current one is consumed (two barriers per B sequence).
// Per 32x32 tile the vector unit spends 16 and-or + 16 max on the row side and 16 and-or + a
// max3 tree on the column side; the cross-lane part of the row maximum is done once per
// (A tile, B sequence) by transposing the 32x32 keys through LDS so that a lane owns a row.
#include "pclip_common.h"

#define FD 128
#define ROWB 256

struct FilipFastArgs {
    const uint16_t* hnA; const uint32_t* mwA; int Br, TAp;
    const uint16_t* hnB; const uint32_t* mwB; int Bc, TBp;
    const float* t;
    float* simA; float* simB;
    uint16_t* argA; uint16_t* argB;
    int jchunk, ngroups;
};

#define FL_IMG 0
#define FL_COLP 65536                          // int   [2 isel][8 at][2 h][256 b]   column partial maxima
#define FL_TR (FL_COLP + 2 * 8 * 2 * 256 * 4)  // int   [8 waves][32 rows][36]       per-wave transpose pad
#define FL_ROWV (FL_TR + 8 * 32 * 36 * 4)      // float [2 isel][256 a]              row maxima (as scores)
#define FL_RED (FL_ROWV + 2 * 256 * 4)         // float [2 kinds][8 waves]
#define FL_MWA (FL_RED + 64)                   // u32   [2][8]   mask words of the two A sequences
#define FL_MWB (FL_MWA + 64)                   // u32   [MAXJ][8] mask words of this workgroup's B sequences
#define FL_MAXJ 64
#define FL_TOTAL (FL_MWB + FL_MAXJ * 32)

#define NEG_KEY ((int)0xFF7FFFFF)  // bits of -FLT_MAX

__device__ __forceinline__ int and_or(int x, int m, int o) { return (x & m) | o; }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

__global__ __launch_bounds__(512, 2) void filip_sim_unit_kernel(FilipFastArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* colp = reinterpret_cast<int*>(smem + FL_COLP);
    float* rowv = reinterpret_cast<float*>(smem + FL_ROWV);
    float* red = reinterpret_cast<float*>(smem + FL_RED);
    uint32_t* mwa_l = reinterpret_cast<uint32_t*>(smem + FL_MWA);
    uint32_t* mwb_l = reinterpret_cast<uint32_t*>(smem + FL_MWB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave id, provably uniform
    const int c = lane & 31, h = lane >> 5;
    const int grp = blockIdx.x % p.ngroups;
    const int chunk = blockIdx.x / p.ngroups;
    const int i0 = grp * 2;
    const int j_begin = chunk * p.jchunk, j_end = min(p.Bc, j_begin + p.jchunk);
    const int ntA = p.TAp >> 5, ntB = p.TBp >> 5;
    const float inv_t = 1.f / fabsf(p.t[0]);
    int* tr = reinterpret_cast<int*>(smem + FL_TR) + w * 32 * 36;

    // Mask words live in LDS for the whole kernel: every later read is an LDS read, so no
    // vector-memory wait inside the main loop can drain the image prefetch.
    if (tid < 16) {
        const int isel = tid >> 3, at = tid & 7;
        mwa_l[tid] = (i0 + isel < p.Br && at < ntA) ? p.mwA[(size_t)(i0 + isel) * ntA + at] : 0u;
    }
    for (int k = tid; k < (j_end - j_begin) * 8; k += 512) {
        const int jj = k >> 3, tt = k & 7;
        mwb_l[k] = (tt < ntB) ? p.mwB[(size_t)(j_begin + jj) * ntB + tt] : 0u;
    }
    __syncthreads();
#define MWA(isel, at) ((uint32_t)__builtin_amdgcn_readfirstlane((int)mwa_l[(isel) * 8 + (at)]))
#define MWB(j, tt) ((uint32_t)__builtin_amdgcn_readfirstlane((int)mwb_l[((j) - j_begin) * 8 + (tt)]))

    // ---- deal the valid A tiles of the two sequences to the waves -------------------------
    int nA[2] = {0, 0};
    int slot_isel[2] = {0, 0}, slot_at[2] = {0, 0};
    uint32_t wA[2] = {0u, 0u};
    {
        int item = 0;
        for (int isel = 0; isel < 2; ++isel)
            for (int at = 0; at < 8; ++at) {
                const uint32_t wd = MWA(isel, at);
                nA[isel] += __popc(wd);
                if (wd) {
                    if ((item & 7) == w) {
                        const int sl = item >> 3;
                        slot_isel[sl] = isel; slot_at[sl] = at; wA[sl] = wd;
                    }
                    ++item;
                }
            }
    }
    bf16x8 afr[2][8];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const uint16_t* src = p.hnA + ((size_t)min(i0 + slot_isel[sl], p.Br - 1) * p.TAp + slot_at[sl] * 32 + c) * FD + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) afr[sl][s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
    }
    // Retire the A-fragment loads HERE: an opaque use makes the compiler wait for them before
    // the main loop; otherwise it places its waits at their first use inside the loop, where
    // (the counter being in-order) they would also drain every image prefetch issued since.
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(afr[sl][s]));

    // ---- image staging: 512 threads x 8 x 16 B = one 64-KB image ------------------------------
    // (straight-line code with named registers: an array captured by a lambda ends up in
    // scratch memory, which would turn the prefetch into a round trip through it)
    const int img_chunks = p.TBp * 16;  // 16-byte chunks of one image
    uint4 pre0, pre1, pre2, pre3, pre4, pre5, pre6, pre7;
#define PREFETCH(jj)                                                                                       \
    {                                                                                                      \
        const uint4* src_ = reinterpret_cast<const uint4*>(p.hnB + (size_t)(jj) * p.TBp * FD);             \
        pre0 = src_[min(tid, img_chunks - 1)];          pre1 = src_[min(tid + 512, img_chunks - 1)];       \
        pre2 = src_[min(tid + 1024, img_chunks - 1)];   pre3 = src_[min(tid + 1536, img_chunks - 1)];      \
        pre4 = src_[min(tid + 2048, img_chunks - 1)];   pre5 = src_[min(tid + 2560, img_chunks - 1)];      \
        pre6 = src_[min(tid + 3072, img_chunks - 1)];   pre7 = src_[min(tid + 3584, img_chunks - 1)];      \
    }
#define COMMIT1(q, v)                                                                                      \
    {                                                                                                      \
        const int g_ = tid + 512 * (q);                                                                    \
        const int row_ = g_ >> 4, ch_ = g_ & 15;                                                           \
        if (g_ < img_chunks) *reinterpret_cast<uint4*>(smem + FL_IMG + row_ * ROWB + ((ch_ ^ (row_ & 15)) << 4)) = v; \
    }
#define COMMIT()                                                                                           \
    {                                                                                                      \
        COMMIT1(0, pre0) COMMIT1(1, pre1) COMMIT1(2, pre2) COMMIT1(3, pre3)                                \
        COMMIT1(4, pre4) COMMIT1(5, pre5) COMMIT1(6, pre6) COMMIT1(7, pre7)                                \
    }
    if (j_begin < j_end) {
        PREFETCH(j_begin)
        COMMIT()
    }
    __syncthreads();

    for (int j = j_begin; j < j_end; ++j) {
        const bool more = j + 1 < j_end;
        if (more) PREFETCH(j + 1)

        // ---- this wave's A tiles against every valid B tile ---------------------------------
        int KM[2][16];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int r = 0; r < 16; ++r) KM[sl][r] = NEG_KEY;
        int nB = 0;
        for (int bt = 0; bt < ntB; ++bt) {
            const uint32_t wB = MWB(j, bt);
            nB += __popc(wB);
            if (wB == 0 || (wA[0] | wA[1]) == 0) continue;
            bf16x8 bfr[8];
            {
                const unsigned char* base = smem + FL_IMG + (bt * 32 + c) * ROWB;
#pragma unroll
                for (int s = 0; s < 8; ++s) bfr[s] = *reinterpret_cast<const bf16x8*>(base + (((2 * s + h) ^ (c & 15)) << 4));
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (wA[sl] == 0) continue;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 2.0f;
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[sl][s], bfr[s], acc, 0, 0, 0);
                int bits[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bits[r] = __float_as_int(acc[r]);
                if (wA[sl] != 0xffffffffu || wB != 0xffffffffu) {  // a partial tile: mask pairs with padding
                    const uint32_t rowbits = wA[sl] >> (4 * h);
                    const bool colok = (wB >> c) & 1u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool ok = colok && ((rowbits >> ((r & 3) + 8 * (r >> 2))) & 1u);
                        bits[r] = ok ? bits[r] : NEG_KEY;
                    }
                }
                int ck[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    KM[sl][r] = max(KM[sl][r], and_or(bits[r], ~7, bt));
                    ck[r] = and_or(bits[r], ~15, r);
                }
                const int m0 = max3i(ck[0], ck[1], ck[2]), m1 = max3i(ck[3], ck[4], ck[5]), m2 = max3i(ck[6], ck[7], ck[8]);
                const int m3 = max3i(ck[9], ck[10], ck[11]), m4 = max3i(ck[12], ck[13], ck[14]);
                const int cbest = max3i(max3i(m0, m1, m2), max3i(m3, m4, ck[15]), NEG_KEY);
                colp[((slot_isel[sl] * 8 + slot_at[sl]) * 2 + h) * 256 + bt * 32 + c] = cbest;
            }
        }

        // ---- rows: transpose the 32x32 keys of each tile through LDS; lane (row, half) scans 16 ----
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (wA[sl] == 0) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[acc_row(r, h) * 36 + c] = KM[sl][r];
            __builtin_amdgcn_wave_barrier();
            int k16[16];
            {
                const int4* rp = reinterpret_cast<const int4*>(tr + c * 36 + h * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int4 v = rp[q];
                    k16[4 * q] = v.x; k16[4 * q + 1] = v.y; k16[4 * q + 2] = v.z; k16[4 * q + 3] = v.w;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int a0 = max3i(k16[0], k16[1], k16[2]), a1 = max3i(k16[3], k16[4], k16[5]), a2 = max3i(k16[6], k16[7], k16[8]);
            const int a3 = max3i(k16[9], k16[10], k16[11]), a4 = max3i(k16[12], k16[13], k16[14]);
            int m = max(max3i(a0, a1, a2), max3i(a3, a4, k16[15]));
            int cc = 15;
#pragma unroll
            for (int k = 14; k >= 0; --k) cc = (k16[k] == m) ? k : cc;
            cc += 16 * h;
            {
                const int om = __shfl_xor(m, 32), oc = __shfl_xor(cc, 32);
                if (om > m || (om == m && oc < cc)) { m = om; cc = oc; }
            }
            // every lane of both halves now holds row `c` of the tile: key m, winning lane cc
            if (h == 0) {
                const bool valid = (wA[sl] >> c) & 1u;
                const bool none = m < 0;
                const float val = none ? -FLT_MAX : __int_as_float(and_or(m, ~7, 4)) - 2.0f;
                rowv[slot_isel[sl] * 256 + slot_at[sl] * 32 + c] = valid ? val : 0.f;
                if (p.argA)
                    p.argA[((size_t)(i0 + slot_isel[sl]) * p.Bc + j) * p.TAp + slot_at[sl] * 32 + c] =
                        (none || !valid) ? (uint16_t)0 : (uint16_t)((m & 7) * 32 + cc);
            }
        }
        __syncthreads();  // #1: image consumed, colp / rowv complete

        if (more) COMMIT()

        // ---- thread (isel, x): column x of the pair and row x of the pair -----------------------
        {
            const int isel = tid >> 8, x = tid & 255;
            const bool live_i = i0 + isel < p.Br;
            float cv = 0.f, rv = 0.f;
            if (live_i && x < p.TBp) {
                const uint32_t wB = mwb_l[(j - j_begin) * 8 + (x >> 5)];
                const bool valid = (wB >> (x & 31)) & 1u;
                int best = (int)0x80000000, barg = 0;
                if (valid) {
                    for (int at = 0; at < ntA; ++at) {
                        if (mwa_l[isel * 8 + at] == 0) continue;
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int k = colp[((isel * 8 + at) * 2 + hh) * 256 + x];
                            if (k > best) { best = k; barg = at * 32 + acc_row(k & 15, hh); }
                        }
                    }
                    cv = (best < 0) ? -FLT_MAX : __int_as_float(and_or(best, ~15, 8)) - 2.0f;
                    if (best < 0) barg = 0;
                }
                if (p.argB) p.argB[((size_t)(i0 + isel) * p.Bc + j) * p.TBp + x] = (uint16_t)barg;
            }
            if (live_i && x < p.TAp && mwa_l[isel * 8 + (x >> 5)] != 0) rv = rowv[isel * 256 + x];
            cv = wave_sum(cv);
            rv = wave_sum(rv);
            if (lane == 0) { red[w] = cv; red[8 + w] = rv; }
        }
        __syncthreads();  // #2: next image in place; colp / rowv free; red complete
        if ((tid & 255) == 0 && i0 + (tid >> 8) < p.Br) {
            const int isel = tid >> 8;
            const float sb = red[isel * 4] + red[isel * 4 + 1] + red[isel * 4 + 2] + red[isel * 4 + 3];
            const float sa = red[8 + isel * 4] + red[8 + isel * 4 + 1] + red[8 + isel * 4 + 2] + red[8 + isel * 4 + 3];
            p.simA[(size_t)(i0 + isel) * p.Bc + j] = sa / fmaxf((float)nA[isel], 1e-6f) * inv_t;
            p.simB[(size_t)(i0 + isel) * p.Bc + j] = sb / fmaxf((float)nB, 1e-6f) * inv_t;
        }
    }
}

int filip_sim_unit_launch(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TAp, const uint16_t* hnB, const uint32_t* maskwB, int Bc,
                          int TBp, const float* t, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB, void* stream) {
    FilipFastArgs a;
    a.hnA = hnA; a.mwA = maskwA; a.Br = Br; a.TAp = TAp;
    a.hnB = hnB; a.mwB = maskwB; a.Bc = Bc; a.TBp = TBp;
    a.t = t; a.simA = sim_A; a.simB = sim_B; a.argA = argA; a.argB = argB;
    a.ngroups = (Br + 1) / 2;
    int nchunk = (1024 + a.ngroups - 1) / a.ngroups;   // about four workgroups per CU over the launch
    int jchunk = (Bc + nchunk - 1) / nchunk;
    if (jchunk < 8) jchunk = Bc < 8 ? Bc : 8;
    if (jchunk > FL_MAXJ) jchunk = FL_MAXJ;
    nchunk = (Bc + jchunk - 1) / jchunk;
    a.jchunk = jchunk;
    static bool lds_set = false;
    if (!lds_set) {
        if (hipFuncSetAttribute((const void*)filip_sim_unit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FL_TOTAL) != hipSuccess)
            return pclip_fail(PCLIP_ELAUNCH, "filip_sim_unit_kernel: cannot raise the dynamic LDS limit");
        lds_set = true;
    }
    hipLaunchKernelGGL(filip_sim_unit_kernel, dim3(a.ngroups * nchunk), dim3(512), FL_TOTAL, (hipStream_t)stream, a);
    return pclip_check_launch("filip_sim_unit_kernel");
}
