// Backward of the FILIP scores for sequences of at most 256 tokens (autograd of
// modules/models.py:332-351), own-stationary form, gfx950.
//
// The max routes every gradient to one (a, b) token pair, so the gradient of an "own" token is a
// weighted sum of partner token rows: the row it chose in every partner sequence, plus the rows of
// the partner tokens that chose it.  Served from L2, these 256-byte row gathers are bound by the L2
// request rate (every partner row is fetched once per own token that uses it: about 50 times the
// bytes of the images).  Here a workgroup keeps the gradients of TWO own sequences in registers
// (64 groups of 16 lanes x 4 + 4 tokens x 128 columns of f32) and streams the partner sequences
// through LDS: each partner image is fetched once per workgroup by LDS-DMA (a contiguous copy,
// double-buffered) and the rows are then gathered from LDS, 16 lanes per row.
//
// What to add where comes from a per-(own, partner) RECORD built by filip_pair_prep_kernel (one
// wave per sequence pair, both directions at once).  Own tokens are addressed by POSITION
// p = (a % 64) * 4 + a / 64: lane group r holds the tokens r, r + 64, r + 128, r + 192 of BOTH
// sequences, i.e. positions 4r .. 4r + 3 of each (ragged sequences and sequences of different
// lengths then spread evenly over the groups, which meet at a barrier after every partner).
//   header     w1 = weight of an own token's chosen row, w2 = weight of the rows of tokens that chose it
//   own[256]   u16: the partner token chosen by the own token at position p (the index of an all-zero
//              row for padding); bit 15: that token chose it back (its row then carries w1 + w2 and
//              is left out of the lists)
//   first[256] u16: the first partner token of p's list (the zero row if the list is empty)
//   cnt[256]   u16: length of p's list
//   seg[256]   u16: start of p's list
//   list[]     u8: partner tokens sorted by the position of the own token they chose
// So every position has exactly two row additions with a fixed place in the instruction stream
// (chosen row, first chooser); longer lists (one position in ten) are finished by a short loop.
#include "pclip_common.h"

#define FD 128
#define ROWB 256   // bytes of a bf16 token row

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ---- record layout ---------------------------------------------------------------------
#define NPOS 256
#define REC_OWN 16
#define REC_FIRST (16 + 2 * NPOS)
#define REC_CNT (16 + 4 * NPOS)
#define REC_SEG (16 + 6 * NPOS)
#define REC_LIST (16 + 8 * NPOS)
__host__ __device__ __forceinline__ int rec_bytes(int TPp) { return REC_LIST + TPp; }
__device__ __forceinline__ int tok_pos(int a) { return (a & 63) * 4 + (a >> 6); }

struct PairPrepArgs {
    const uint16_t* argA; const uint16_t* argB;   // [pair][TAp], [pair][TBp], pair = i*Bc + j
    const uint32_t* mwA; const uint32_t* mwB;
    const float* gA; const float* gB;             // [pair]
    const float* icA; const float* icB;           // 1 / valid tokens (0 for an empty sequence)
    const float* t;
    int Br, Bc, TAp, TBp;
    unsigned char* recA;   // [Br][Bc] records of the A-side pass (own = A tokens), or NULL
    unsigned char* recB;   // [Bc][Br] records of the B-side pass (own = B tokens), or NULL
};

// One direction of one pair: own[] / par[] are the sanitised arg rows in LDS (0xffff = padding).
__device__ __forceinline__ void build_record(const uint16_t* own, int TOp, const uint16_t* par, int TPp, int* hist, unsigned char* rec,
                                             float w1, float w2, int lane) {
    uint16_t* r_own = reinterpret_cast<uint16_t*>(rec + REC_OWN);
    uint16_t* r_first = reinterpret_cast<uint16_t*>(rec + REC_FIRST);
    uint16_t* r_cnt = reinterpret_cast<uint16_t*>(rec + REC_CNT);
    uint16_t* r_seg = reinterpret_cast<uint16_t*>(rec + REC_SEG);
    unsigned char* r_list = rec + REC_LIST;
    if (lane == 0) {
        reinterpret_cast<float*>(rec)[0] = w1;
        reinterpret_cast<float*>(rec)[1] = w2;
        reinterpret_cast<uint32_t*>(rec)[2] = 0u;
        reinterpret_cast<uint32_t*>(rec)[3] = 0u;
    }
    for (int q = lane; q < NPOS; q += 64) {   // q: position; its token a = (q % 4) * 64 + q / 4
        const int a = (q & 3) * 64 + (q >> 2);
        int ent = TPp;                        // the zero row
        if (a < TOp) {
            const int x = own[a];
            if (x != 0xffff) ent = (x < TPp ? x : TPp - 1) | ((x < TPp && (int)par[x] == a) ? 0x8000 : 0);
        }
        r_own[q] = (uint16_t)ent;
        hist[q] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    // partner tokens that chose an own token which did not choose them back: counted per position
    for (int b = lane; b < TPp; b += 64) {
        const int y = par[b];
        if (y != 0xffff && y < TOp && (int)own[y] != b) atomicAdd(&hist[tok_pos(y)], 1);
    }
    __builtin_amdgcn_wave_barrier();
    {   // exclusive scan, four consecutive positions per lane
        int c[4];
        const int p0 = 4 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = hist[p0 + k];
        const int s = c[0] + c[1] + c[2] + c[3];
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(incl, o);
            if (lane >= o) incl += n;
        }
        int run = incl - s;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hist[p0 + k] = run;
            r_seg[p0 + k] = (uint16_t)run;
            r_cnt[p0 + k] = (uint16_t)c[k];
            run += c[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int b = lane; b < TPp; b += 64) {
        const int y = par[b];
        if (y != 0xffff && y < TOp && (int)own[y] != b) {
            const int pos = atomicAdd(&hist[tok_pos(y)], 1);
            r_list[pos] = (unsigned char)b;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int q = lane; q < NPOS; q += 64) r_first[q] = r_cnt[q] ? (uint16_t)r_list[r_seg[q]] : (uint16_t)TPp;
    __builtin_amdgcn_wave_barrier();
}

// One wave per sequence pair.  Dynamic LDS per wave: both arg rows (u16), counters (int, the
// 256 positions), both records.
__global__ __launch_bounds__(256) void filip_pair_prep_kernel(PairPrepArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t pair = (size_t)blockIdx.x * 4 + w;
    if (pair >= (size_t)p.Br * p.Bc) return;
    const int i = (int)(pair / p.Bc), j = (int)(pair - (size_t)i * p.Bc);
    const int TAp = p.TAp, TBp = p.TBp;
    const int rbA = rec_bytes(TBp), rbB = rec_bytes(TAp);
    unsigned char* base = sm + (size_t)w * (2 * (TAp + TBp) + 4 * NPOS + rbA + rbB);
    uint16_t* a_l = reinterpret_cast<uint16_t*>(base);
    uint16_t* b_l = a_l + TAp;
    int* hist = reinterpret_cast<int*>(b_l + TBp);
    unsigned char* rA = reinterpret_cast<unsigned char*>(hist + NPOS);
    unsigned char* rB = rA + rbA;
    const uint16_t* ga = p.argA + pair * TAp;
    const uint16_t* gb = p.argB + pair * TBp;
    const uint32_t* ma = p.mwA + (size_t)i * (TAp >> 5);
    const uint32_t* mb = p.mwB + (size_t)j * (TBp >> 5);
    // entries of padding tokens are unspecified: they enter as an index that matches nothing
    for (int k = 4 * lane; k < TAp; k += 256) {
        const uint2 v = *reinterpret_cast<const uint2*>(ga + k);
        const uint32_t m4 = ma[k >> 5] >> (k & 31);
        const uint32_t lo = ((m4 & 1u) ? (v.x & 0xffffu) : 0xffffu) | ((m4 & 2u) ? (v.x & 0xffff0000u) : 0xffff0000u);
        const uint32_t hi = ((m4 & 4u) ? (v.y & 0xffffu) : 0xffffu) | ((m4 & 8u) ? (v.y & 0xffff0000u) : 0xffff0000u);
        *reinterpret_cast<uint2*>(a_l + k) = make_uint2(lo, hi);
    }
    for (int k = 4 * lane; k < TBp; k += 256) {
        const uint2 v = *reinterpret_cast<const uint2*>(gb + k);
        const uint32_t m4 = mb[k >> 5] >> (k & 31);
        const uint32_t lo = ((m4 & 1u) ? (v.x & 0xffffu) : 0xffffu) | ((m4 & 2u) ? (v.x & 0xffff0000u) : 0xffff0000u);
        const uint32_t hi = ((m4 & 4u) ? (v.y & 0xffffu) : 0xffffu) | ((m4 & 8u) ? (v.y & 0xffff0000u) : 0xffff0000u);
        *reinterpret_cast<uint2*>(b_l + k) = make_uint2(lo, hi);
    }
    __builtin_amdgcn_wave_barrier();
    const float icA = p.icA[i], icB = p.icB[j], inv_t = 1.f / fabsf(p.t[0]);
    const float ga_ij = p.gA[pair], gb_ij = p.gB[pair];
    const bool live = icA > 0.f && icB > 0.f;   // a pair with an empty side routes nothing
    if (p.recA) {
        build_record(a_l, TAp, b_l, TBp, hist, rA, live ? ga_ij * icA * inv_t : 0.f, live ? gb_ij * icB * inv_t : 0.f, lane);
        uint4* dst = reinterpret_cast<uint4*>(p.recA + ((size_t)i * p.Bc + j) * rbA);
        for (int k = lane; k < (rbA >> 4); k += 64) dst[k] = reinterpret_cast<const uint4*>(rA)[k];
    }
    if (p.recB) {
        build_record(b_l, TBp, a_l, TAp, hist, rB, live ? gb_ij * icB * inv_t : 0.f, live ? ga_ij * icA * inv_t : 0.f, lane);
        uint4* dst = reinterpret_cast<uint4*>(p.recB + ((size_t)j * p.Br + i) * rbB);
        for (int k = lane; k < (rbB >> 4); k += 64) dst[k] = reinterpret_cast<const uint4*>(rB)[k];
    }
}

// ---- the gather ------------------------------------------------------------------------
struct GatherLdsArgs {
    const uint16_t* hnP;        // partner images [Np][TPp][128] bf16
    const uint32_t* mwO;        // own mask words [No][TOp/32]
    const uint32_t* mwP;        // partner mask words [Np][TPp/32]
    const unsigned char* rec;   // [No][Np] records
    float* dn;                  // [No*TOp][nsplit][128] f32 out
    int No, TOp, Np, TPp, nsplit, qper, recb;
};

#ifdef BWD_STAMPS   // diagnostic build: shader cycles per segment of the step loop, summed over all waves
__device__ unsigned long long g_bwd_dbg[8];
extern "C" void pclip_debug_read(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_dbg), sizeof(g_bwd_dbg));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_dbg), z, sizeof(z));
}
#define STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); seg_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

// 1 KiB of LDS-DMA: lane l copies 16 bytes from its own global address to lds_dst + 16 l.
__device__ __forceinline__ void dma_1k(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);   // wave-uniform by construction
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// Workgroup = 64 groups of 16 lanes; group r holds the gradient rows of the tokens r + 64k (k < 4)
// of sequence 2g in slots 0-3 and of sequence 2g + 1 in slots 4-7 (g = blockIdx / nsplit), lane l
// of the group the columns 8l .. 8l + 7.  blockIdx % nsplit selects the partner range (blocks that
// share an XCD then share it, so the XCD's L2 serves the image copies of its 32 workgroups).
// LDS: two partner images, each followed by one all-zero row | two pairs of records | copy
// lengths of the partners of the range.
// F16: the partner images are f16 copies of the bf16 images (exact for unit-norm rows down to
// 7.6e-6, 3e-8 absolute below), so that conversion and multiply-add are one v_fma_mix_f32.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ void row_fma(float (&a)[8], const uint4& d, float w) {
    if constexpr (F16) {
        // a += w * f16 half of a data dword, converted inside the instruction (hipcc emits a separate
        // v_cvt_f32_f16 per element for the plain C form)
#define MIX_LO(acc_, dw_) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc_) : "v"(dw_), "v"(w))
#define MIX_HI(acc_, dw_) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc_) : "v"(dw_), "v"(w))
        MIX_LO(a[0], d.x); MIX_HI(a[1], d.x);
        MIX_LO(a[2], d.y); MIX_HI(a[3], d.y);
        MIX_LO(a[4], d.z); MIX_HI(a[5], d.z);
        MIX_LO(a[6], d.w); MIX_HI(a[7], d.w);
#undef MIX_LO
#undef MIX_HI
    } else {
        a[0] = fmaf(w, __uint_as_float(d.x << 16), a[0]);
        a[1] = fmaf(w, __uint_as_float(d.x & 0xffff0000u), a[1]);
        a[2] = fmaf(w, __uint_as_float(d.y << 16), a[2]);
        a[3] = fmaf(w, __uint_as_float(d.y & 0xffff0000u), a[3]);
        a[4] = fmaf(w, __uint_as_float(d.z << 16), a[4]);
        a[5] = fmaf(w, __uint_as_float(d.z & 0xffff0000u), a[5]);
        a[6] = fmaf(w, __uint_as_float(d.w << 16), a[6]);
        a[7] = fmaf(w, __uint_as_float(d.w & 0xffff0000u), a[7]);
    }
}
#define ENT16(v, k) (((k) & 1) ? ((&(v).x)[(k) >> 1] >> 16) : ((&(v).x)[(k) >> 1] & 0xffffu))

template <bool F16>
__global__ __launch_bounds__(1024) void filip_gather_lds_kernel(GatherLdsArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, l16 = tid & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x % p.nsplit, grp = blockIdx.x / p.nsplit;
    const int q_lo = split * p.qper;
    const int nq = max(min(p.Np, q_lo + p.qper) - q_lo, 0);
    const int IMG = p.TPp * ROWB, IMGB = IMG + ROWB;
    const int META = (2 * p.recb + 1023) & ~1023;
    unsigned char* meta = sm + 2 * IMGB;
    unsigned char* plen = meta + 2 * META;
    const int r = tid >> 4;                       // lane group 0..63
    const int o0 = 2 * grp;
    const bool two = o0 + 1 < p.No;
    const int o1 = two ? o0 + 1 : o0;
    const int ntP = p.TPp >> 5;
    const int TOp = p.TOp;

    // Slots some group of this wave has a valid token in (wave-uniform): the others are skipped.
    uint32_t wmask = 0u;
    {
        uint32_t vm = 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int a = r + 64 * (k & 3), o = (k < 4) ? o0 : o1;
            if (a < TOp && (k < 4 || two)) vm |= ((p.mwO[(size_t)o * (TOp >> 5) + (a >> 5)] >> (a & 31)) & 1u) << k;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) wmask |= __any((vm >> k) & 1u) ? (1u << k) : 0u;
    }
    wmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)wmask);

    if (tid < 32) reinterpret_cast<uint4*>(sm + (tid >> 4) * IMGB + IMG)[tid & 15] = make_uint4(0u, 0u, 0u, 0u);   // the zero rows
    // 1-KiB pieces to copy per partner: up to its last tile with a valid token
    for (int x = tid; x < nq; x += 1024) {
        int last = -1;
        for (int tt = 0; tt < ntP; ++tt)
            if (p.mwP[(size_t)(q_lo + x) * ntP + tt]) last = tt;
        plen[x] = (unsigned char)((last + 1) * 8);
    }
    __syncthreads();

    const uint32_t lds_img = (uint32_t)(uintptr_t)(lds_ptr_t)sm;
    const uint32_t lds_meta = (uint32_t)(uintptr_t)(lds_ptr_t)meta;
    const unsigned char* hnP_bytes = reinterpret_cast<const unsigned char*>(p.hnP);
    const int nmp = META >> 10;
    // Copies of step s into buffer buf are cut into five parts per wave (four image pieces, one record
    // piece), issued between the row sums of the step before so that the copy engine, the LDS reads
    // and the vector pipe are busy at the same time rather than one after the other.
    auto issue_part = [&](int s, int buf, int part) {
        const int q = q_lo + s;
        if (part < 4) {
            const int np = __builtin_amdgcn_readfirstlane((int)plen[s]);
            const int c = w + 16 * part;
            if (c < np) dma_1k(hnP_bytes + (size_t)q * IMG + lane * 16 + c * 1024, lds_img + (uint32_t)(buf * IMGB + c * 1024));
        } else if (w < nmp) {   // the two records, as one run of 16-byte pieces
            const int v = w * 1024 + lane * 16;
            const unsigned char* r0 = p.rec + ((size_t)o0 * p.Np + q) * p.recb;
            const unsigned char* r1 = p.rec + ((size_t)o1 * p.Np + q) * p.recb;
            const unsigned char* msrc = v < p.recb ? r0 + v : (v < 2 * p.recb ? r1 + (v - p.recb) : r0);
            dma_1k(msrc, lds_meta + (uint32_t)(buf * META + w * 1024));
        }
    };
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int part = 0; part < 5; ++part) issue_part(s, buf, part);
    };

    float acc[8][8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;

    if (nq > 0) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#ifdef BWD_STAMPS
    unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < nq; ++s) {
        const int cb = s & 1;
        STAMP(0);
        const bool more_steps = s + 1 < nq;
        bool issued = false;
        STAMP(1);
#ifndef ABL_NOCOMPUTE
        {
            const unsigned char* R0 = meta + cb * META;
            const unsigned char* R1 = R0 + p.recb;
            const float w1a = __int_as_float(__builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(R0)[0]));
            const float w2a = __int_as_float(__builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(R0)[1]));
            const float w1b = __int_as_float(__builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(R1)[0]));
            const float w2b = __int_as_float(__builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(R1)[1]));
            // slots with work in this step: a pair with an empty side carries zero weights
            const uint32_t act = wmask & (((w1a != 0.f || w2a != 0.f) ? 0x0fu : 0u) | ((two && (w1b != 0.f || w2b != 0.f)) ? 0xf0u : 0u));
            if (act) {
                const float w12a = w1a + w2a, w12b = w1b + w2b;
                uint4 own8, fst8, cnt8, seg8;
                {
                    const uint2 a0 = *reinterpret_cast<const uint2*>(R0 + REC_OWN + 8 * r), a1 = *reinterpret_cast<const uint2*>(R1 + REC_OWN + 8 * r);
                    const uint2 f0 = *reinterpret_cast<const uint2*>(R0 + REC_FIRST + 8 * r), f1 = *reinterpret_cast<const uint2*>(R1 + REC_FIRST + 8 * r);
                    const uint2 c0 = *reinterpret_cast<const uint2*>(R0 + REC_CNT + 8 * r), c1 = *reinterpret_cast<const uint2*>(R1 + REC_CNT + 8 * r);
                    const uint2 s0 = *reinterpret_cast<const uint2*>(R0 + REC_SEG + 8 * r), s1 = *reinterpret_cast<const uint2*>(R1 + REC_SEG + 8 * r);
                    own8 = make_uint4(a0.x, a0.y, a1.x, a1.y);
                    fst8 = make_uint4(f0.x, f0.y, f1.x, f1.y);
                    cnt8 = make_uint4(c0.x, c0.y, c1.x, c1.y);
                    seg8 = make_uint4(s0.x, s0.y, s1.x, s1.y);
                }
                const unsigned char* im = sm + cb * IMGB + l16 * 16;
                // second list entries (where a list has two or more), requested ahead of the fixed rows
                uint32_t sec[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned char* lst = (k < 4 ? R0 : R1) + REC_LIST;
                    sec[k] = (ENT16(cnt8, k) > 1u) ? (uint32_t)lst[ENT16(seg8, k) + 1u] : (uint32_t)p.TPp;
                }
                // one slot (two rows) at a time, the next slot's rows requested before this one's sums
                uint4 ra[2], rf[2];
#define LOAD_TOK(buf_, k)                                                                        \
    do {                                                                                          \
        ra[buf_] = *reinterpret_cast<const uint4*>(im + (ENT16(own8, k) & 0x1ffu) * ROWB);       \
        rf[buf_] = *reinterpret_cast<const uint4*>(im + ENT16(fst8, k) * ROWB);                  \
    } while (0)
#define SUM_TOK(buf_, k)                                                                         \
    do {                                                                                          \
        row_fma<F16>(acc[k], ra[buf_], (ENT16(own8, k) & 0x8000u) ? ((k) < 4 ? w12a : w12b) : ((k) < 4 ? w1a : w1b)); \
        row_fma<F16>(acc[k], rf[buf_], (k) < 4 ? w2a : w2b);                                      \
    } while (0)
                issued = true;
#pragma unroll
                for (int k = 0; k <= 8; ++k) {
                    if (k < 8 && ((act >> (k & 7)) & 1u)) LOAD_TOK(k & 1, k & 7);
                    if (k < 5 && more_steps) issue_part(s + 1, cb ^ 1, k);
                    if (k > 0 && ((act >> (k - 1)) & 1u)) SUM_TOK((k - 1) & 1, (k - 1) & 7);
                }
#undef LOAD_TOK
#undef SUM_TOK
#ifdef BWD_STAMPS
                asm volatile("" :: "v"(acc[0][0]), "v"(acc[7][7]));
#endif
                STAMP(2);
                // second entries, then whatever is left of longer lists
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t c = ENT16(cnt8, k);
                    if (((act >> k) & 1u) && __any(c > 1u)) {
                        const float w2 = k < 4 ? w2a : w2b;
                        const uint4 d = *reinterpret_cast<const uint4*>(im + sec[k] * ROWB);
                        row_fma<F16>(acc[k], d, w2);
                        if (__any(c > 2u)) {
                            const unsigned char* lst = (k < 4 ? R0 : R1) + REC_LIST + ENT16(seg8, k);
                            for (uint32_t e = 2; e < c; ++e) {
                                const uint4 d2 = *reinterpret_cast<const uint4*>(im + (uint32_t)lst[e] * ROWB);
                                row_fma<F16>(acc[k], d2, w2);
                            }
                        }
                    }
                }
            }
        }
#endif
        if (!issued && more_steps) issue(s + 1, cb ^ 1);
        STAMP(3);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        STAMP(4);
        __builtin_amdgcn_s_barrier();
        STAMP(5);
    }
#ifdef BWD_STAMPS
    if (lane == 0)
        for (int k = 0; k < 6; ++k) atomicAdd(&g_bwd_dbg[k], seg_[k]);
    if (tid == 0) atomicAdd(&g_bwd_dbg[7], (unsigned long long)nq);
#endif

#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int a = r + 64 * (k & 3);
        if (a < TOp && (k < 4 || two)) {
            const int o = k < 4 ? o0 : o1;
            float4* out = reinterpret_cast<float4*>(p.dn + (((size_t)o * TOp + a) * p.nsplit + split) * FD + l16 * 8);
            out[0] = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
            out[1] = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
        }
    }
}

// bf16 -> f16, eight elements per thread (only for images whose rows have norm <= 1)
__global__ __launch_bounds__(256) void bf16_to_f16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 d = src[i];
    half8_t h;
    h[0] = (_Float16)__uint_as_float(d.x << 16); h[1] = (_Float16)__uint_as_float(d.x & 0xffff0000u);
    h[2] = (_Float16)__uint_as_float(d.y << 16); h[3] = (_Float16)__uint_as_float(d.y & 0xffff0000u);
    h[4] = (_Float16)__uint_as_float(d.z << 16); h[5] = (_Float16)__uint_as_float(d.z & 0xffff0000u);
    h[6] = (_Float16)__uint_as_float(d.w << 16); h[7] = (_Float16)__uint_as_float(d.w & 0xffff0000u);
    dst[i] = __builtin_bit_cast(uint4, h);
}

// ---- host side -------------------------------------------------------------------------
static inline int tp32(int T) { return (T + 31) / 32 * 32; }

// Partner ranges per own group: enough workgroups to cover the 256 CUs once, ranges of at least
// eight partners, a power of two (so that blocks sharing an XCD share a range).
static int fast_nsplit(int No, int Np) {
    const int groups = (No + 1) / 2;
    int ns = 1;
    while (groups * ns < 256 && ns < 8 && Np / (2 * ns) >= 8) ns *= 2;
    return ns;
}

struct FastWs {
    size_t recA, recB, icA, icB, img16A, img16B, partA, partB, total, dnA, dnB, total_fused;
    int nsA, nsB, rbA, rbB;
};
static FastWs fast_layout(int Br, int TA, int Bc, int TB) {
    const size_t TAp = tp32(TA), TBp = tp32(TB);
    FastWs L;
    L.nsA = fast_nsplit(Br, Bc);
    L.nsB = fast_nsplit(Bc, Br);
    L.rbA = rec_bytes((int)TBp);   // A-side pass: own = A tokens, lists hold B tokens
    L.rbB = rec_bytes((int)TAp);
    size_t off = 0;
    L.recA = off; off += pclip_align_up((size_t)Br * Bc * L.rbA + 1024, 256);
    L.recB = off; off += pclip_align_up((size_t)Br * Bc * L.rbB + 1024, 256);
    L.icA = off;  off += pclip_align_up((size_t)Br * 4, 256);
    L.icB = off;  off += pclip_align_up((size_t)Bc * 4, 256);
    L.img16A = off; off += pclip_align_up((size_t)Br * TAp * ROWB, 256);   // f16 copies of the images (unit-norm operands)
    L.img16B = off; off += pclip_align_up((size_t)Bc * TBp * ROWB, 256);
    L.partA = off; off += L.nsA > 1 ? pclip_align_up((size_t)L.nsA * Br * TAp * FD * 4, 256) : 0;
    L.partB = off; off += L.nsB > 1 ? pclip_align_up((size_t)L.nsB * Bc * TBp * FD * 4, 256) : 0;
    L.total = off;
    L.dnA = off; off += L.nsA > 1 ? 0 : pclip_align_up((size_t)Br * TAp * FD * 4, 256);
    L.dnB = off; off += L.nsB > 1 ? 0 : pclip_align_up((size_t)Bc * TBp * FD * 4, 256);
    L.total_fused = off;
    return L;
}

bool filip_bwd_fast_ok(int TA, int TB) { return tp32(TA) <= 256 && tp32(TB) <= 256; }
size_t filip_bwd_fast_ic_offset(int Br, int TA, int Bc, int TB, int side) {
    const FastWs L = fast_layout(Br, TA, Bc, TB);
    return side == 0 ? L.icA : L.icB;
}
size_t filip_bwd_fast_ws_bytes(int Br, int TA, int Bc, int TB, int fused) {
    const FastWs L = fast_layout(Br, TA, Bc, TB);
    return fused ? L.total_fused : L.total;
}

// After the pre kernel has filled icA / icB (and dt): records, then per requested side the gather
// and what follows it (nothing, the sum of the range partials, or the prep backward over them).
int filip_bwd_fast_run(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA, const uint16_t* hnB, const uint32_t* maskwB, int Bc, int TB,
                       const float* t, int unit_norm, const uint16_t* argA, const uint16_t* argB, const float* gA, const float* gB, float* dnA,
                       float* dnB,
                       const FilipBwdTail& tailA, const FilipBwdTail& tailB, void* ws, size_t ws_bytes, void* stream) {
    const FastWs L = fast_layout(Br, TA, Bc, TB);
    const bool fused = tailA.h || tailB.h;
    PCLIP_REQUIRE(ws_bytes >= (fused ? L.total_fused : L.total), "filip backward: workspace %zu < %zu bytes", ws_bytes,
                  fused ? L.total_fused : L.total);
    const int TAp = tp32(TA), TBp = tp32(TB);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    const bool doA = dnA || tailA.h, doB = dnB || tailB.h;
    {
        PairPrepArgs q;
        q.argA = argA; q.argB = argB; q.mwA = maskwA; q.mwB = maskwB; q.gA = gA; q.gB = gB;
        q.icA = (const float*)(w + L.icA); q.icB = (const float*)(w + L.icB); q.t = t;
        q.Br = Br; q.Bc = Bc; q.TAp = TAp; q.TBp = TBp;
        q.recA = doA ? (unsigned char*)(w + L.recA) : nullptr;
        q.recB = doB ? (unsigned char*)(w + L.recB) : nullptr;
        const size_t lds = (size_t)4 * (2 * (TAp + TBp) + 4 * NPOS + L.rbA + L.rbB);
        static PclipLdsLimit lim;
        if (int rc = pclip_ensure_lds(lim, (const void*)filip_pair_prep_kernel, lds, "filip_pair_prep_kernel")) return rc;
        const size_t npair = (size_t)Br * Bc;
        hipLaunchKernelGGL(filip_pair_prep_kernel, dim3((unsigned)((npair + 3) / 4)), dim3(256), lds, st, q);
        if (int rc = pclip_check_launch("filip_pair_prep_kernel")) return rc;
    }
    for (int side = 0; side < 2; ++side) {
        if (!(side == 0 ? doA : doB)) continue;
        GatherLdsArgs a;
        const FilipBwdTail& tail = side == 0 ? tailA : tailB;
        float* dn = side == 0 ? dnA : dnB;
        int ns;
        if (unit_norm) {   // f16 copy of this side's partner images
            const uint16_t* src = side == 0 ? hnB : hnA;
            uint16_t* dst = (uint16_t*)(w + (side == 0 ? L.img16B : L.img16A));
            const size_t n8 = (side == 0 ? (size_t)Bc * TBp : (size_t)Br * TAp) * (FD / 8);
            hipLaunchKernelGGL(bf16_to_f16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n8);
            if (side == 0) hnB = dst; else hnA = dst;
        }
        if (side == 0) {
            a.hnP = hnB; a.mwO = maskwA; a.mwP = maskwB; a.rec = (const unsigned char*)(w + L.recA);
            a.No = Br; a.TOp = TAp; a.Np = Bc; a.TPp = TBp; a.recb = L.rbA; ns = L.nsA;
        } else {
            a.hnP = hnA; a.mwO = maskwB; a.mwP = maskwA; a.rec = (const unsigned char*)(w + L.recB);
            a.No = Bc; a.TOp = TBp; a.Np = Br; a.TPp = TAp; a.recb = L.rbB; ns = L.nsB;
        }
        a.nsplit = ns;
        a.qper = (a.Np + ns - 1) / ns;
        float* part = (float*)(w + (side == 0 ? L.partA : L.partB));
        float* scratch_dn = (float*)(w + (side == 0 ? L.dnA : L.dnB));
        a.dn = ns > 1 ? part : (tail.h ? scratch_dn : dn);
        const int META = (2 * a.recb + 1023) & ~1023;
        const size_t lds = (size_t)2 * (a.TPp + 1) * ROWB + 2 * META + pclip_align_up((size_t)a.qper, 16);
        static PclipLdsLimit lim16, lim;
        if (unit_norm) {
            if (int rc = pclip_ensure_lds(lim16, (const void*)filip_gather_lds_kernel<true>, lds, "filip_gather_lds_kernel")) return rc;
            hipLaunchKernelGGL(filip_gather_lds_kernel<true>, dim3((unsigned)(((a.No + 1) / 2) * ns)), dim3(1024), lds, st, a);
        } else {
            if (int rc = pclip_ensure_lds(lim, (const void*)filip_gather_lds_kernel<false>, lds, "filip_gather_lds_kernel")) return rc;
            hipLaunchKernelGGL(filip_gather_lds_kernel<false>, dim3((unsigned)(((a.No + 1) / 2) * ns)), dim3(1024), lds, st, a);
        }
        if (int rc = pclip_check_launch("filip_gather_lds_kernel")) return rc;
        if (tail.h) {
            if (int rc = filip_prep_bwd_launch(tail.h, a.dn, ns, FD, (size_t)ns * FD, a.mwO, a.No, tail.T, tail.d, tail.normalise, tail.tsign, 0,
                                               tail.dh, stream))
                return rc;
        } else if (ns > 1) {
            if (int rc = filip_window_sum_launch(part, a.No, a.TOp, ns, dn, a.mwO, stream)) return rc;
        }
    }
    return PCLIP_OK;
}
