// FILIP score kernel for UNIT-NORM operands, streaming form (gfx950).
//
// Work decomposition.  A workgroup of 16 waves (the only one on its CU) owns two A sequences;
// their valid 32-token tiles (at most 16) are dealt one per wave (so the four SIMDs carry
// ceil(n/4) tiles each) and stay in registers as MFMA A operands for the whole kernel.  The B
// side is a STREAM of valid 32-token tiles — the tiles of this workgroup's chunk of B sequences
// back to back, all-padding tiles left out — that flows through a ring of tile slots (8 KiB
// each) in LDS, filled by LDS-DMA (global_load_lds_dwordx4, 1-KiB pieces) several tiles ahead
// of the matrix cores.  A tile's image in LDS is [k-step 8][lane half 2][token 32] chunks of
// 16 B — the DMA's per-lane source address does the reshaping — so a lane reads its eight MFMA
// B fragments from ONE address register with immediate offsets, conflict-free.  One workgroup
// barrier per epoch of SE tiles is the only synchronisation of the main loop.
//
// Per 32x32 tile: 8 v_mfma_f32_32x32x16_bf16 with the accumulator started at 4.0: |a.b| <= 1,
// so every similarity becomes a float in [3,5], whose sign and upper seven exponent bits are
// the same for all of them.  Shifted left by seven the bit pattern still orders like the value
// (the lowest exponent bit and all 23 mantissa bits stay) and the low seven bits are free for
// the arg index: 3 bits of B tile for the row maxima, 4 bits of accumulator register for the
// column maxima, ONE key per element for both sides (one v_lshl_or_b32).  Row and column
// maxima are integer maxima of these keys; nothing of the value is dropped, a maximum differs
// from the exact one only by the f32 rounding of the accumulation near 4 (ulp 2.4e-7 / 4.8e-7).
// Padding columns get a negative key; padding rows of an A tile are fed as copies of a valid row (no
// masking, the winner is mapped back).  Column maxima of the A tiles held by different waves
// meet in LDS through a
// 64-bit integer atomic max (key in the high word, A tile and lane half in the low word).
// Rows finish once per (A tile, B sequence) by transposing the 32x32 keys through a per-wave
// LDS pad; columns finish once per epoch, after its barrier, by a rotating subset of waves.
#include "pclip_common.h"

// Diagnostic build (-DSTREAM_STAMPS): every wave adds up the shader-clock cycles it spends in
// eight segments of the main loop and stores them to g_stream_dbg[(block*16 + wave)*8 + k].
#ifdef STREAM_STAMPS
extern "C" { unsigned long long* g_stream_dbg = nullptr; }
#define STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); seg_[k] += (unsigned)(now_ - last_); last_ = now_; } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
// Diagnostic build (-DSTREAM_TIMELINE): wall-clock stamps (100 MHz counter) of every workgroup's phases and the CU it
// ran on, to g_stream_tl[block * 32 + k]: 0 entry, 1 pairing done, 2 set-up done, 3 first epoch landed, 4 + w end of
// wave w's main loop, 20 end of the workgroup, 21 hardware id, 22 XCC id, 23 / 24 the shader-clock counter at entry and
// end (with 0 and 20: the clock the workgroup's CU really ran at).
#ifdef STREAM_TIMELINE
extern "C" { unsigned long long* g_stream_tl = nullptr; }
#define TL(k) do { if (p.tl && lane == 0 && w == 0) p.tl[(size_t)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TL(k) do {} while (0)
#endif
#ifndef STREAM_WGS
#define STREAM_WGS 768       // workgroups aimed at per launch (three per CU; one fills a CU's LDS).  A workgroup's set-up costs
                             // about 9 us and nothing overlaps it, but the pairs differ in cost (3 or 4 tile rounds at ragged
                             // lengths) and three or four workgroups per CU even that out: benchmark shape 0.700 / 0.690 /
                             // 0.690 ms with 512 / 768 / 1024, every token valid 0.957 / 0.967 / 0.97; counts that are not a
                             // multiple of the 256 CUs lose 4-8 %
#endif
#ifndef ABL_MFMA_STEPS
#define ABL_MFMA_STEPS 8   // timing experiments only: fewer k-steps (wrong results)
#endif
#define FD 128
#define TILE_BYTES 8192
#define SMAXJ 64              // B sequences per workgroup at most

#define NEG_KEY ((int)0xFF7FFFFF)              // bits of -FLT_MAX
#define NEG_KEY64 ((long long)0x8000000000000000LL)

#define FLAG_LAST 1u
#define FLAG_NULL 2u

struct FilipStreamArgs {
    const uint16_t* hnA; const uint32_t* mwA; int Br, TAp;
    const uint16_t* hnB; const uint32_t* mwB; int Bc, TBp;
    const float* t;
    float* simA; float* simB;
    uint16_t* argA; uint16_t* argB;
    int jchunk, ngroups, nchunk, cpw;
    int ldc, col0;   // the outputs are [Br, ldc(, T)] arrays of which this launch fills the columns col0 .. col0 + Bc - 1
    // Sequences longer than 256 tokens arrive cut into SEGMENTS of 256 (filip_long.hip): "sequence" g of this kernel is then
    // segment g % seg of sequence g / seg, whose image starts 256 (g % seg) token rows into that sequence's `rows` rows
    // (B side: 8 (g % seg) tiles into its `tiles` tiles), and its mask words are eight of the sequence's.  seg = 1: whole sequences.
    int segA, rowsA, segB, tilesB;
    int plain;   // segments: block b is (pair b % ngroups, chunk b / ngroups), no ranking of the pairs by cost (with few, similar
                 // pairs the ranked dealing leaves some shader engines twice the work of others)
    float* rowval; float* colval;   // optional [Br, ldc, 256]: the maximum behind every argA / argB entry (unscaled; -FLT_MAX: none)
#ifdef STREAM_STAMPS
    unsigned long long* dbg;
#endif
#ifdef STREAM_TIMELINE
    unsigned long long* tl;
#endif
};

// Geometry: 16 waves (four per SIMD, 128 registers each) with one A tile each; epochs of four
// tiles (longer epochs amortise the per-epoch barrier, DMA issue and duties: 4-5 % faster than
// two), two epochs in the ring.
struct StreamCfg {
    static constexpr int NW = 16;
#ifndef STREAM_SE
#define STREAM_SE 4
#define STREAM_SNE 2
#endif
    static constexpr int SE = STREAM_SE;               // tiles per epoch (one barrier each); even
    static constexpr int SNE = STREAM_SNE;             // epochs held by the ring
    static constexpr int SR = SE * SNE;                // ring slots
    static constexpr int SMAXT = SMAXJ * 8 + SE;
    static constexpr int S_RING = 0;
    static constexpr int S_TR = SR * TILE_BYTES;                 // int   [NW][32][36]
    static constexpr int S_COL = S_TR + NW * 32 * 36 * 4;        // i64   [2 par][SE][2 isel][32 c]
    static constexpr int S_ROWP = S_COL + 2 * SE * 2 * 32 * 8;   // float [3 par][SE][16 A tiles]
    static constexpr int S_COLP = S_ROWP + 3 * SE * 16 * 4;      // float [3 par][SE][2 isel]
    static constexpr int S_MWA = S_COLP + 3 * SE * 2 * 4;        // u32   [2][8]
    static constexpr int S_CNT = S_MWA + 64;                     // int   [4]: stream length
    static constexpr int S_SCL = S_CNT + 16;                     // float [4]: 1/|t|, 1/nA[0], 1/nA[1]
    static constexpr int S_NB = S_SCL + 16;                      // int   [SMAXJ] valid tokens of each B sequence
    static constexpr int S_META = S_NB + SMAXJ * 4;              // uint4 [SMAXT]
    static constexpr int S_TOTAL = S_META + SMAXT * 16;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ int and_or_i(int x, int m, int o) { return (x & m) | o; }
__device__ __forceinline__ int max3_i(int a, int b, int c) { return max(max(a, b), c); }

// sum over each 16-lane row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
#define DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
    DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    DPP_ADD(0x141);  // row_half_mirror
    DPP_ADD(0x140);  // row_mirror
#undef DPP_ADD
    return v;
}
// the value of lane l ^ 16 / l ^ 32, through the gfx950 swap instructions (no LDS round trip)
__device__ __forceinline__ int xor16_i(int v, int lane) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (lane & 16) ? (int)r[0] : (int)r[1];
}
__device__ __forceinline__ int xor32_i(int v, int lane) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (lane & 32) ? (int)r[0] : (int)r[1];
}
// sum over each 32-lane half, result in every lane of the half
__device__ __forceinline__ float half32_sum(float v, int lane) {
    v = row16_sum(v);
    return v + __int_as_float(xor16_i(__float_as_int(v), lane));
}

// One finished accumulator tile -> its 16 keys per lane, in place of the accumulator:
// key = bits(4 + a.b) << 7 | B tile << 4 | register.  When the B tile has padding columns,
// mask_keys gives the lanes on them (bit l of colmask clear) NEG_KEY, which leaves the running
// row maxima alone (their column result is never read).
#define KEY_SHIFT 7
#define KEY_TOP 0x40000000   // the seven bits the shift drops: sign 0, exponent 100 0000x
__device__ __forceinline__ float key_value(int key) { return __int_as_float((key >> KEY_SHIFT) | KEY_TOP) - 4.0f; }

__device__ __forceinline__ void make_keys(const f32x16& acc, int (&key)[16], int bt4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int tag;   // a scalar register per tag, so that shift and tag are ONE v_lshl_or_b32
        asm("s_or_b32 %0, %1, %2" : "=s"(tag) : "s"(bt4), "n"(r) : "scc");
        key[r] = (int)(((uint32_t)__float_as_int(acc[r]) << KEY_SHIFT) | (uint32_t)tag);
    }
}
__device__ __forceinline__ void mask_keys(int (&key)[16], unsigned long long colmask) {
#pragma unroll
    for (int r = 0; r < 16; ++r) key[r] = __builtin_amdgcn_inverse_ballot_w64(colmask) ? key[r] : NEG_KEY;
}

// Row side.  FRESH: the first product of a B sequence overwrites the running maxima, which saves
// both the sixteen maxima here and a reset of the sixteen registers after every sequence.
template <bool FRESH>
__device__ __forceinline__ void row_fold(const int (&key)[16], int (&KM)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) KM[r] = FRESH ? key[r] : max(KM[r], key[r]);
}

// Column side: a tree of max3, depth three (padding rows of the A tile hold copies of a valid row).
__device__ __forceinline__ int col_max(const int (&key)[16]) {
    const int a0 = max3_i(key[0], key[1], key[2]), a1 = max3_i(key[3], key[4], key[5]), a2 = max3_i(key[6], key[7], key[8]);
    const int a3 = max3_i(key[9], key[10], key[11]), a4 = max3_i(key[12], key[13], key[14]);
    return max3_i(max3_i(a0, a1, a2), max3_i(a3, a4, key[15]), NEG_KEY);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N < 0, "add the wait count");
}

__global__ __launch_bounds__(1024) void filip_sim_stream_kernel(FilipStreamArgs p) {
    using C = StreamCfg;
    constexpr int NW = C::NW, SE = C::SE, SNE = C::SNE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long long* col64 = reinterpret_cast<long long*>(smem + C::S_COL);
    float* rowp = reinterpret_cast<float*>(smem + C::S_ROWP);
    float* colpart = reinterpret_cast<float*>(smem + C::S_COLP);
    uint32_t* mwa_l = reinterpret_cast<uint32_t*>(smem + C::S_MWA);
    int* cnt_l = reinterpret_cast<int*>(smem + C::S_CNT);
    float* scl_l = reinterpret_cast<float*>(smem + C::S_SCL);
    int* nb_l = reinterpret_cast<int*>(smem + C::S_NB);
    uint4* meta = reinterpret_cast<uint4*>(smem + C::S_META);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    // Which (pair of A sequences, chunk of B sequences) this block takes.  The hardware deals the blocks of a launch
    // round-robin and in order over 8 XCDs x 4 shader engines (block b goes to engine b % 32) and only fills a free CU
    // of the engine whose turn it is, so an engine's share of the work is the blocks with its residue, and they start
    // in index order.  With the pairing below the workgroups differ in cost (tile rounds of the pair); the pairs are
    // therefore ranked by cost and dealt over the 32 engines in snake order (every engine gets the same mix), all
    // chunks of an engine's heaviest pair first (longest first: the short ones fill the end).
    int grp = blockIdx.x % p.ngroups;
    int chunk = blockIdx.x / p.ngroups;
    int srank = -1;
#ifndef ABL_NOPAIR
    if (p.Br <= NW * 64 && !p.plain) {
        // an engine's list: chunks in blocks of `cpw` (what its 8 CUs run at the same time stays within few chunks of B,
        // which its XCD's L2 then holds), inside a block the pairs heaviest first
        const int r = blockIdx.x & 31, k = blockIdx.x >> 5;
        const int rows = (p.ngroups + 31) >> 5, cpw = p.cpw;
        const int per_block = rows * cpw;
        const int cb = k / per_block, k2 = k - cb * per_block;
        const int j = k2 / cpw;
        chunk = cb * cpw + (k2 - j * cpw);
        srank = 32 * j + ((j & 1) ? 31 - r : r);
        if (srank >= p.ngroups || chunk >= p.nchunk) return;
    }
#endif
    const int j_begin = chunk * p.jchunk, j_end = min(p.Bc, j_begin + p.jchunk);
    const int nj = j_end - j_begin;
    const int ntA = p.TAp >> 5, ntB = p.TBp >> 5;
    TL(0);
#ifdef STREAM_TIMELINE
    if (p.tl && lane == 0 && w == 0) p.tl[(size_t)blockIdx.x * 32 + 23] = __builtin_readcyclecounter();
#endif

    // ---- which two A sequences: pairs whose valid-tile counts sum to a multiple of four --------
    // Sixteen waves sit on four SIMDs, so a workgroup takes ceil(tiles / 4) tile rounds per B
    // tile: 13 tiles cost as much as 16.  Every workgroup derives the same pairing from the mask
    // words alone (ranks by index order inside the classes tiles mod 4; class 0 pairs with
    // itself, 2 with itself, 1 with 3; what is left over pairs up in index order).  Batches
    // beyond one sequence per thread keep the neighbours (2g, 2g + 1).
    int iseq[2] = {grp * 2, grp * 2 + 1 < p.Br ? grp * 2 + 1 : -1};
#ifndef ABL_NOPAIR
    if (p.Br <= NW * 64) {
        int* wtot = reinterpret_cast<int*>(smem + C::S_RING);                   // [NW][4] class counts per wave
        uint16_t* pairmem = reinterpret_cast<uint16_t*>(smem + C::S_RING + 1024);  // [ngroups][2]
        int* ctot = reinterpret_cast<int*>(smem + C::S_RING + 256);                // [NW][8] cost counts per wave
        uint8_t* ntmem = smem + C::S_RING + 3072;                                   // [Br] valid tiles of a sequence
        uint16_t* sortedmem = reinterpret_cast<uint16_t*>(smem + C::S_RING + 4096);  // [ngroups] pairs, heaviest first
        int cls = -1;
        if (tid < p.Br) {
            int nt = 0;
            const int ntr = p.rowsA >> 5, s8 = (tid % p.segA) * 8;   // (segment tid % segA of sequence tid / segA)
            for (int k = 0; k < ntA && s8 + k < ntr; ++k) nt += p.mwA[(size_t)(tid / p.segA) * ntr + s8 + k] != 0u;
            cls = nt & 3;
            ntmem[tid] = (uint8_t)nt;
        }
        int rank = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned long long m = __ballot(cls == r);
            if (cls == r) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wtot[w * 4 + r] = __popcll(m);
        }
        for (int k = tid; k < 2 * p.ngroups; k += NW * 64) pairmem[k] = 0xFFFFu;
        __syncthreads();
        int n[4] = {0, 0, 0, 0};
        for (int k = 0; k < NW; ++k) {
            const int4 v = *reinterpret_cast<const int4*>(wtot + k * 4);
            if (k < w && cls >= 0) rank += cls == 0 ? v.x : cls == 1 ? v.y : cls == 2 ? v.z : v.w;
            n[0] += v.x; n[1] += v.y; n[2] += v.z; n[3] += v.w;
        }
        if (cls >= 0) {
            const int p0 = n[0] >> 1, p2 = n[2] >> 1, m13 = min(n[1], n[3]);
            int pid, pos, pool = -1;
            if (cls == 0)      { if (rank < 2 * p0) { pid = rank >> 1; pos = rank & 1; } else pool = 0; }
            else if (cls == 2) { if (rank < 2 * p2) { pid = p0 + (rank >> 1); pos = rank & 1; } else pool = n[0] & 1; }
            else               { if (rank < m13) { pid = p0 + p2 + rank; pos = cls == 3; } else pool = (n[0] & 1) + (n[2] & 1) + rank - m13; }
            if (pool >= 0) { pid = p0 + p2 + m13 + (pool >> 1); pos = pool & 1; }
            pairmem[2 * pid + pos] = (uint16_t)tid;
        }
        __syncthreads();
        if (!p.plain) {   // rank the pairs by cost = tile rounds of their workgroups (ceil(tiles / 4)), heaviest first, ties in index order
            int cost = -1;
            if (tid < p.ngroups) {
                const int m0 = pairmem[2 * tid], m1 = pairmem[2 * tid + 1];
                cost = ((m0 != 0xFFFF ? (int)ntmem[m0] : 0) + (m1 != 0xFFFF ? (int)ntmem[m1] : 0) + 3) >> 2;
            }
            int crank = 0;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const unsigned long long m = __ballot(cost == 4 - v);
                if (cost == 4 - v) crank = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == 0) ctot[w * 8 + v] = __popcll(m);
            }
            __syncthreads();
            if (cost >= 0) {
                const int v = 4 - cost;
                int base = 0;
                const int nwv = (p.ngroups + 63) >> 6;   // waves that hold pairs
                for (int k = 0; k < nwv; ++k)
#pragma unroll
                    for (int u = 0; u < 5; ++u)
                        if (u < v || (u == v && k < w)) base += ctot[k * 8 + u];
                sortedmem[base + crank] = (uint16_t)tid;
            }
            __syncthreads();
            grp = sortedmem[srank];
        }
        const int a0 = pairmem[2 * grp], a1 = pairmem[2 * grp + 1];
        iseq[0] = a0 == 0xFFFF ? -1 : a0;
        iseq[1] = a1 == 0xFFFF ? -1 : a1;
        if (iseq[0] < 0) { iseq[0] = iseq[1]; iseq[1] = -1; }
        __syncthreads();   // the ring region is handed to the LDS-DMA below
    }
#endif
    iseq[0] = __builtin_amdgcn_readfirstlane(iseq[0]);
    iseq[1] = __builtin_amdgcn_readfirstlane(iseq[1]);
    TL(1);
#define ISEQ(isel) ((isel) ? iseq[1] : iseq[0])

    // ---- set-up: A mask words, column-key array, the B tile stream ---------------------------
    if (tid < 16) {
        const int isel = tid >> 3, at = tid & 7;
        uint32_t wd = 0u;
        if (ISEQ(isel) >= 0 && at < ntA) {
            // segment ISEQ % segA of sequence ISEQ / segA: its tiles 8 (ISEQ % segA) + at, none behind the sequence's last
            const int tile = (ISEQ(isel) % p.segA) * 8 + at, nt = p.rowsA >> 5;
            if (tile < nt) wd = p.mwA[(size_t)(ISEQ(isel) / p.segA) * nt + tile];
        }
        mwa_l[tid] = wd;
    }
    for (int k = tid; k < 2 * SE * 2 * 32; k += NW * 64) col64[k] = NEG_KEY64;
    if (w == 1) {  // one wave lays out the stream: lane = local B sequence
        const int jl = lane;
        uint32_t wd[8];
        int nvt = 0, ntok = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            wd[k] = 0u;
            if (jl < nj && k < ntB) {
                const int tile = ((j_begin + jl) % p.segB) * 8 + k;
                if (tile < p.tilesB) wd[k] = p.mwB[(size_t)((j_begin + jl) / p.segB) * p.tilesB + tile];
            }
            nvt += wd[k] != 0;
            ntok += __popc(wd[k]);
        }
        const int jg = j_begin + jl;   // first tile of the sequence's image among all 32-token tiles of hnB
        const uint32_t tile0 = p.segB == 1 ? (uint32_t)(jg * p.tilesB) : (uint32_t)((jg / p.segB) * p.tilesB + (jg % p.segB) * 8);
        const int mine = jl < nj ? max(nvt, 1) : 0;  // an empty sequence still gets one (empty) entry
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(incl, o);
            if (lane >= o) incl += n;
        }
        int pos = incl - mine;
        if (jl < nj) {
            nb_l[jl] = ntok;
            if (nvt == 0) {
                meta[pos] = make_uint4(0u, 0u, (uint32_t)jl | (FLAG_LAST << 16), FLAG_LAST);
            } else {
                int seen = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (wd[k]) {
                        ++seen;
                        meta[pos++] = make_uint4(wd[k], tile0 + (uint32_t)k, (uint32_t)(jl | (k << 8)) | ((seen == nvt ? FLAG_LAST : 0u) << 16),
                                                 seen == nvt ? FLAG_LAST : 0u);
                    }
            }
        }
        const int total = __shfl(incl, 63);
        const int padded = (total + SE - 1) / SE * SE;
        if (lane < padded - total) meta[total + lane] = make_uint4(0u, 0u, FLAG_NULL << 16, FLAG_NULL);
        if (lane == 0) cnt_l[0] = padded;
    }
    __syncthreads();
#define MWA(isel, at) ((uint32_t)__builtin_amdgcn_readfirstlane((int)mwa_l[(isel) * 8 + (at)]))

    // ---- deal the valid A tiles of the two sequences to the waves: item k -> wave k ----------
    int slot_isel = 0, slot_at = 0, n_items = 0;
    uint32_t wA = 0u;
    {
        int item = 0, nA0 = 0, nA1 = 0;
        for (int isel = 0; isel < 2; ++isel)
            for (int at = 0; at < 8; ++at) {
                const uint32_t wd = MWA(isel, at);
                if (isel == 0) nA0 += __popc(wd); else nA1 += __popc(wd);
                if (wd) {
                    if (item == w) { slot_isel = isel; slot_at = at; wA = wd; }
                    ++item;
                }
            }
        n_items = item;
        if (tid == 0) {
            scl_l[0] = 1.f / fabsf(p.t[0]);
            scl_l[1] = 1.f / fmaxf((float)nA0, 1e-6f);
            scl_l[2] = 1.f / fmaxf((float)nA1, 1e-6f);
        }
    }
    bf16x8 afr[8];
    {
        // A padding row of the tile is fed as a COPY of the tile's lowest valid row: its products repeat that row's
        // exactly, so no column maximum changes and no row of the tile has to be masked out of the column side; a
        // column winner that is such a copy is mapped back to the row it copies where the columns are finished.
        const int arow = ((wA >> c) & 1u) ? c : __builtin_ctz(wA | 0x80000000u);
        const int ig = max(ISEQ(slot_isel), 0);
        const size_t row0 = p.segA == 1 ? (size_t)ig * p.rowsA : (size_t)(ig / p.segA) * p.rowsA + (size_t)(ig % p.segA) * 256;
        const uint16_t* src = p.hnA + (row0 + slot_at * 32 + arow) * FD + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) afr[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
    }
    // retire the A-fragment loads here (an opaque use), before any LDS-DMA is in flight
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(afr[s]));

    TL(2);
    const int T = __builtin_amdgcn_readfirstlane(cnt_l[0]);
    const int nE = T / SE;

    // LDS-DMA.  Piece s = w % 8 of a tile is its k-step s: lane l (half l / 32, token l % 32)
    // fetches bytes [32 s + 16 half, +16) of the token's 256-byte row and lands at chunk
    // 64 s + l of the tile image.  Wave w serves the tiles w / 8, w / 8 + 2, ... of every epoch.
    // Issued through inline asm: hipcc orders every later LDS read behind a DMA it knows of
    // (vmcnt(0)), which would serialise the ring; this way the copies are counted by hand (SE / 2
    // per epoch and wave, retired by the vmcnt waits in front of each epoch barrier).
    const int piece = w & 7, dq = w >> 3;
    const unsigned char* hnB_bytes = reinterpret_cast<const unsigned char*>(p.hnB);
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + C::S_RING) + (uint32_t)piece * 1024u;
    // the lane's part of the source address: token ln % 32, k-step `piece`, half ln / 32
    const uint32_t dsrc_off = (uint32_t)(c * (FD * 2) + 32 * piece + 16 * h);
    // g[k]: index of the tile (among all 32-token tiles of hnB) that wave-local DMA k of the epoch fetches
    auto issue_epoch = [&](int E, const uint32_t (&g)[SE / 2]) {
#pragma unroll
        for (int k = 0; k < SE / 2; ++k) {   // SE * 8 pieces per epoch over 16 waves
            const int tq = dq + 2 * k;
            const unsigned char* src = hnB_bytes + (size_t)g[k] * (32 * FD * 2);   // scalar
            const uint32_t dst = ring_lds + (uint32_t)((E % SNE) * SE + tq) * TILE_BYTES;
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(dsrc_off), "s"(src), "s"(dst) : "memory");
        }
    };
    // The records of an epoch's tiles, read by eight lanes at once (lanes 0-3: the tiles of epoch E0, lanes 4-7:
    // those of epoch E1) and handed out as scalars by v_readlane.
    auto read_records = [&](int E0, int E1) { return meta[((lane & 4) ? E1 : E0) * SE + (lane & 3)]; };
    auto dma_tiles = [&](const uint4& rec, int base_lane, uint32_t (&g)[SE / 2]) {
#pragma unroll
        for (int k = 0; k < SE / 2; ++k) g[k] = (uint32_t)__builtin_amdgcn_readlane((int)rec.y, base_lane + dq + 2 * k);
    };

    uint16_t* const arg_base = p.argA ? p.argA + ((size_t)max(ISEQ(slot_isel), 0) * p.ldc + p.col0 + j_begin) * p.TAp : nullptr;
    float* const rowval_base = p.rowval ? p.rowval + ((size_t)max(ISEQ(slot_isel), 0) * p.ldc + p.col0 + j_begin) * p.TAp : nullptr;
    int KM[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) KM[r] = NEG_KEY;
    bool fresh = true;   // wave-uniform: no product of the current B sequence has been folded into KM yet

    // prologue: epochs 0 .. SNE-2 in flight, epoch 0 landed
    static_assert(SE == 4, "the tile records of an epoch are read by four lanes");
    for (int E = 0; E < SNE - 1; ++E)
        if (E < nE) {
            const uint4 rec = read_records(E, E);
            uint32_t g[SE / 2];
            dma_tiles(rec, 0, g);
            issue_epoch(E, g);
        }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    TL(3);

    // Emits the two scores of every sequence whose last tile lies in epoch E (all of its parts
    // are complete one barrier after that epoch's own).  One wave does it — the last one when it
    // holds no A tile, else the first (which the SIMD serves first) — with the LDS reads spread
    // over its lanes: lane = (A sequence, tile of the epoch, A tile); lanes 0 and 32 then walk the
    // epoch's tiles in order and keep the running column sum of the sequence being streamed.
    static_assert(SE <= 4, "emit_epoch spreads at most four tiles of an epoch over the lanes");
    float runB = 0.f;
    const int emit_w = n_items < NW ? NW - 1 : 0;
    auto emit_epoch = [&](int E) {
        if (w != emit_w) return;
        const int par = E % 3;
        const int e_isel = lane >> 5, e_q = min((lane >> 3) & 3, SE - 1), e_at = lane & 7;
        float rs = mwa_l[e_isel * 8 + e_at] ? rowp[(par * SE + e_q) * 16 + e_isel * 8 + e_at] : 0.f;
        rs += __shfl_xor(rs, 1);
        rs += __shfl_xor(rs, 2);
        rs += __shfl_xor(rs, 4);
        const float cp = colpart[(par * SE + e_q) * 2 + e_isel];
        const uint4 m = meta[E * SE + e_q];
        const float nbv = fmaxf((float)nb_l[m.z & 0xff], 1e-6f);
        const float sc0 = scl_l[0], scA = scl_l[1 + e_isel];
        const bool head = (lane & 31) == 0 && ISEQ(e_isel) >= 0;
#pragma unroll
        for (int q = 0; q < SE; ++q) {
            const int src = (lane & 32) + q * 8;
            const float rs_q = __shfl(rs, src), cp_q = __shfl(cp, src), nb_q = __shfl(nbv, src);
            const uint32_t fl_q = __shfl(m.w, src), z_q = __shfl(m.z, src);
            if (head && !(fl_q & FLAG_NULL)) {
                runB += cp_q;
                if (fl_q & FLAG_LAST) {
                    const size_t o = (size_t)ISEQ(e_isel) * p.ldc + p.col0 + j_begin + (int)(z_q & 0xff);
                    p.simA[o] = rs_q * scA * sc0;
                    p.simB[o] = runB / nb_q * sc0;
                    runB = 0.f;
                }
            }
        }
    };

#ifdef STREAM_STAMPS
    unsigned seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_readcyclecounter();
#endif
    for (int e = 0; e < nE; ++e) {
        const bool ahead = e + SNE - 1 < nE;
        // this epoch's tile records and the tiles of the epoch to fetch, as scalars
        const uint4 rec = read_records(e, ahead ? e + SNE - 1 : e);
        if (ahead) {
            uint32_t g[SE / 2];
            dma_tiles(rec, 4, g);
            issue_epoch(e + SNE - 1, g);
        }
        STAMP(0);
        const int ring0 = (e % SNE) * SE;
        const int par2 = e & 1, par3 = e % 3;
        uint32_t t_wB[SE], t_mz[SE];   // column mask; B sequence | B tile << 8 | flags << 16
#pragma unroll
        for (int q = 0; q < SE; ++q) {
            t_wB[q] = (uint32_t)__builtin_amdgcn_readlane((int)rec.x, q);
            t_mz[q] = (uint32_t)__builtin_amdgcn_readlane((int)rec.z, q);
        }
        STAMP(1);

#pragma unroll
        for (int q = 0; q < SE; ++q) {
            const uint32_t wB = t_wB[q], mz = t_mz[q], fl = mz >> 16;
            const int bt = (mz >> 8) & 7;
            // Issue priority by progress: of the four waves on a SIMD the one that is furthest behind
            // in the epoch goes first.  Left alone the SIMD serves its oldest wave first, the waves
            // finish the epoch one after the other and the last one runs its products alone, at a
            // third of the issue rate, while the others wait at the barrier (5 % of the kernel).
            if (q == 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            if (wB != 0 && wA != 0) {
                bf16x8 bfr[8];
                {
                    const unsigned char* base = smem + C::S_RING + (ring0 + q) * TILE_BYTES + lane * 16;
#pragma unroll
#ifdef ABL_NOFRAG   // timing experiments only: no fragment reads (wrong results)
                    for (int s = 0; s < 8; ++s) { bfr[s] = afr[(s + 1) & 7]; asm volatile("" : "+v"(bfr[s])); }
                    asm volatile("" ::"v"(base));
#elif defined(ABL_HALFFRAG)   // timing experiments only: half the fragment reads
                    for (int s = 0; s < 4; ++s) bfr[s] = *reinterpret_cast<const bf16x8*>(base + 1024 * s);
                    for (int s = 4; s < 8; ++s) { bfr[s] = bfr[s - 4]; asm volatile("" : "+v"(bfr[s])); }
#else
                    for (int s = 0; s < 8; ++s) bfr[s] = *reinterpret_cast<const bf16x8*>(base + 1024 * s);
#endif
                }
                const f32x16 four = {4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f, 4.f};
                f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], bfr[0], four, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < ABL_MFMA_STEPS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[s], bfr[s], acc, 0, 0, 0);
#ifndef ABL_NOHOIST
                // four fragment reads in flight ahead of the matrix instructions (16 registers):
                // one LDS latency per tile instead of four, 2 % faster
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#endif
#ifdef STREAM_STAMPS
                asm volatile("s_nop 0" ::"v"(acc));   // the matrix results have arrived
                STAMP(2);
#endif
#ifdef ABL_NOEPI
                asm volatile("" ::"v"(acc));
#else
                int key[16];
                make_keys(acc, key, bt << 4);
                if (wB != 0xffffffffu) mask_keys(key, ((unsigned long long)wB << 32) | wB);
                if (fresh) row_fold<true>(key, KM);
                else row_fold<false>(key, KM);
                fresh = false;
                const int cb = col_max(key);
#ifndef ABL_NOATOMIC
                atomicMax(&col64[((par2 * SE + q) * 2 + slot_isel) * 32 + c], ((long long)cb << 32) | (long long)((slot_at << 1) | h));
#else
                asm volatile("" ::"v"(cb));
#endif
#endif
                STAMP(3);
            }
#ifndef ABL_NOROWEPI   // timing experiments only: no row epilogue (wrong results)
            if ((fl & FLAG_LAST) && wA != 0) {
                // ---- rows of the finished B sequence: transpose the keys, lane (row, half) scans 16 ----
                int* tr = reinterpret_cast<int*>(smem + C::S_TR) + w * 32 * 36;
                {
                    // an LDS-space pointer (32 bits): passed through an empty asm so that its arithmetic belongs to this
                    // branch and not to every product — as a generic pointer the sixteen writes became flat stores
                    typedef __attribute__((address_space(3))) int lds_int;
                    lds_int* trw = (lds_int*)(tr + 4 * h * 36 + c);
                    asm volatile("" : "+v"(trw));
#pragma unroll
                    for (int r = 0; r < 16; ++r) trw[((r & 3) + 8 * (r >> 2)) * 36] = KM[r];
                }
                const bool empty_b = fresh;   // an empty B sequence: nothing was folded in, KM is stale
                fresh = true;
                __builtin_amdgcn_wave_barrier();
                int k16[16];
                {
                    const int4* rp = reinterpret_cast<const int4*>(tr + c * 36 + h * 16);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int4 v = rp[q4];
                        k16[4 * q4] = v.x; k16[4 * q4 + 1] = v.y; k16[4 * q4 + 2] = v.z; k16[4 * q4 + 3] = v.w;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // All sixteen keys of a row carry the same register tag in their low four bits: it is
                // replaced by 15 - column, so that the maximum brings its (lowest) column along.
#pragma unroll
                for (int k = 0; k < 16; ++k) k16[k] = and_or_i(k16[k], ~15, 15 - k);
                const int a0 = max3_i(k16[0], k16[1], k16[2]), a1 = max3_i(k16[3], k16[4], k16[5]), a2 = max3_i(k16[6], k16[7], k16[8]);
                const int a3 = max3_i(k16[9], k16[10], k16[11]), a4 = max3_i(k16[12], k16[13], k16[14]);
                int m = max(max3_i(a0, a1, a2), max3_i(a3, a4, k16[15]));
                m = empty_b ? NEG_KEY : m;
                int cc;
                {
                    // the other half's best; on equal keys the lower half (the lower columns) wins
                    const int om = xor32_i(m, lane);
                    const bool take = om + h > m;
                    m = take ? om : m;
                    cc = 15 - (m & 15) + 16 * (take ? 1 - h : h);
                }
                // every lane of both halves now holds row `c` of the tile: key m, winning lane cc
                const bool valid = (wA >> c) & 1u;
                const bool none = m < 0;
                const float val = none ? -FLT_MAX : key_value(m);
                const float rs = half32_sum(valid ? val : 0.f, lane);
                if (lane == 0) rowp[(par3 * SE + q) * 16 + slot_isel * 8 + slot_at] = rs;
                if (h == 0 && p.argA)   // scalar base of this wave's A sequence and B chunk, 32-bit lane offset
                    arg_base[(uint32_t)((int)(mz & 0xff) * p.TAp + slot_at * 32 + c)] =
                        (none || !valid) ? (uint16_t)0 : (uint16_t)(((m >> 4) & 7) * 32 + cc);
                if (h == 0 && rowval_base) rowval_base[(uint32_t)((int)(mz & 0xff) * p.TAp + slot_at * 32 + c)] = val;
                STAMP(4);
            }
#endif
        }

        // epoch e+1 has landed once all but the DMAs of the later epochs are retired
        if (ahead) wait_vmcnt<(SNE - 2) * (SE / 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS atomics and writes are done
        STAMP(5);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(3);   // duty waves start the next epoch behind the others
        STAMP(6);

        // ---- columns of this epoch's tiles, by SE waves: waves without an A tile when there are
        // enough of them, else the oldest one of every SIMD (served first, they have the most slack
        // before the next barrier: 1 % faster than rotating the duty) ------------------------------
        const int duty0 = (NW - n_items >= SE) ? NW - SE : 0;
#ifndef ABL_NODUTY   // timing experiments only: columns never finished (wrong results)
        if (w >= duty0 && w < duty0 + SE) {
            const int q = w - duty0, isel = h;
            const uint4 m4 = meta[e * SE + q];
            long long* cp = &col64[((par2 * SE + q) * 2 + isel) * 32 + c];
            const long long key = *cp;
            *cp = NEG_KEY64;
            if (!(m4.w & FLAG_NULL)) {
                const bool valid = ((m4.x >> c) & 1u) && (ISEQ(isel) >= 0);
                float cv = 0.f;
                if (valid) {
                    const int hi = (int)(key >> 32), lo = (int)(key & 0xffffffffLL);
                    int barg = 0;
                    if (hi < 0) {
                        cv = -FLT_MAX;
                    } else {
                        cv = key_value(hi);
                        const int at = lo >> 1, rr = acc_row(hi & 15, lo & 1);
                        const uint32_t wd = mwa_l[isel * 8 + at];
                        barg = at * 32 + (((wd >> rr) & 1u) ? rr : __builtin_ctz(wd | 0x80000000u));   // a copy -> the row it copies
                    }
                    if (p.argB)
                        p.argB[((size_t)ISEQ(isel) * p.ldc + p.col0 + j_begin + (int)(m4.z & 0xff)) * p.TBp + ((m4.z >> 8) & 7) * 32 + c] = (uint16_t)barg;
                    if (p.colval)
                        p.colval[((size_t)ISEQ(isel) * p.ldc + p.col0 + j_begin + (int)(m4.z & 0xff)) * p.TBp + ((m4.z >> 8) & 7) * 32 + c] = cv;
                }
                const float cs = half32_sum(cv, lane);
                if (c == 0) colpart[(par3 * SE + q) * 2 + isel] = cs;
            }
        }
#endif
        if (e > 0) emit_epoch(e - 1);
        STAMP(7);
    }
#ifdef STREAM_STAMPS
    if (lane == 0 && p.dbg)
        for (int k = 0; k < 8; ++k) p.dbg[((size_t)blockIdx.x * NW + w) * 8 + k] = seg_[k];
#endif
#ifdef STREAM_TIMELINE
    if (p.tl && lane == 0) p.tl[(size_t)blockIdx.x * 32 + 4 + w] = __builtin_amdgcn_s_memrealtime();
#endif
    if (nE > 0) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        emit_epoch(nE - 1);
    }
#ifdef STREAM_TIMELINE
    if (p.tl && lane == 0 && w == 0) {
        p.tl[(size_t)blockIdx.x * 32 + 20] = __builtin_amdgcn_s_memrealtime();
        p.tl[(size_t)blockIdx.x * 32 + 21] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        p.tl[(size_t)blockIdx.x * 32 + 22] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
        p.tl[(size_t)blockIdx.x * 32 + 24] = __builtin_readcyclecounter();
    }
#endif
}

int filip_sim_stream_launch(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TAp, const uint16_t* hnB, const uint32_t* maskwB, int Bc,
                            int TBp, const float* t, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB, int col_offset, int ldc,
                            int wgs_target, void* stream) {
    return filip_sim_stream_launch_seg(hnA, maskwA, Br, TAp, 1, TAp, hnB, maskwB, Bc, TBp, 1, TBp >> 5, t, sim_A, sim_B, argA, argB, nullptr, nullptr,
                                       col_offset, ldc, wgs_target, stream);
}

int filip_sim_stream_launch_seg(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TAp, int segA, int rowsA, const uint16_t* hnB,
                                const uint32_t* maskwB, int Bc, int TBp, int segB, int tilesB, const float* t, float* sim_A, float* sim_B,
                                uint16_t* argA, uint16_t* argB, float* rowval, float* colval, int col_offset, int ldc, int wgs_target,
                                void* stream) {
    using C = StreamCfg;
    FilipStreamArgs a;
    a.hnA = hnA; a.mwA = maskwA; a.Br = Br; a.TAp = TAp;
    a.hnB = hnB; a.mwB = maskwB; a.Bc = Bc; a.TBp = TBp;
    a.t = t; a.simA = sim_A; a.simB = sim_B; a.argA = argA; a.argB = argB;
    a.ngroups = (Br + 1) / 2;
    a.ldc = ldc; a.col0 = col_offset;
    a.segA = segA; a.rowsA = rowsA; a.segB = segB; a.tilesB = tilesB; a.rowval = rowval; a.colval = colval;
    a.plain = segA > 1 || segB > 1;
#ifdef STREAM_STAMPS
    a.dbg = g_stream_dbg;
#endif
#ifdef STREAM_TIMELINE
    a.tl = g_stream_tl;
#endif
    if (wgs_target < 1 || wgs_target > 65536) wgs_target = STREAM_WGS;
    int nchunk = (wgs_target + a.ngroups - 1) / a.ngroups;   // about two workgroups per CU over the launch
    int jchunk = (Bc + nchunk - 1) / nchunk;
    if (jchunk < 8) jchunk = Bc < 8 ? Bc : 8;
    if (jchunk > SMAXJ) jchunk = SMAXJ;
    if (segA > 1 || segB > 1) {
        // Segments of long sequences: few, nearly full "sequences" (8 tiles each).  The workgroups of a launch run in
        // rounds of one per compute unit, so the chunk length is the one with the shortest estimated time
        // rounds x (tile steps of a chunk + a workgroup's set-up, about nine tile steps), fewer workgroups on a tie.
        const long cus = 256;
        long best = -1;
        for (int jc = 1; jc <= SMAXJ && jc <= Bc; ++jc) {
            const long nwg = (long)a.ngroups * ((Bc + jc - 1) / jc);
            const long est = ((nwg + cus - 1) / cus) * (8L * jc + 9);
            if (best < 0 || est < best) { best = est; jchunk = jc; }   // ascending: the longest chunk wins a tie last -> use <=
            else if (est == best) jchunk = jc;
        }
    }
    nchunk = (Bc + jchunk - 1) / jchunk;
    a.jchunk = jchunk;
    a.nchunk = nchunk;
    static PclipLdsLimit lds_limit;
    if (int rc = pclip_ensure_lds(lds_limit, (const void*)filip_sim_stream_kernel, C::S_TOTAL, "filip_sim_stream_kernel")) return rc;
    // blocks: one per (pair, chunk); with the cost-ranked dealing the pairs are padded to whole rows of 32 (see the kernel)
    unsigned nblocks = (unsigned)(a.ngroups * nchunk);
    a.cpw = 1;
#ifndef ABL_NOPAIR
    if (Br <= C::NW * 64 && !a.plain) {
        const int rows = (a.ngroups + 31) / 32;
        a.cpw = rows >= 8 ? 1 : 8 / rows;                                  // chunks that an engine's 8 CUs work on together
        const int nblk = (nchunk + a.cpw - 1) / a.cpw;                     // chunk blocks (the last one may be short)
        nblocks = (unsigned)(32 * rows * a.cpw * nblk);
    }
#endif
    hipLaunchKernelGGL(filip_sim_stream_kernel, dim3(nblocks), dim3(C::NW * 64), C::S_TOTAL, (hipStream_t)stream, a);
    return pclip_check_launch("filip_sim_stream_kernel");
}
