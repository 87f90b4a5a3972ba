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
// windows of about 4 MB of padded images (about 3 MB of valid rows with ragged lengths); a
// gather launch covers four consecutive windows, with workgroup b on window b % 4 — blocks b
// and b + 8 are dispatched to the same XCD, so each XCD's L2 only ever sees one window's images
// during a launch, and XCDs k and k + 4 share a window, each taking every other 16-token chunk
// of the own sequences — and writes (first launch) or adds to (later launches, for partner
// sets beyond four windows) one of four PARTIAL gradients; a last pass sums the partials.
// (Eight windows of 2 MB, one XCD each, hit the L2 a little more often but write and re-read
// twice the partials: gather 306 -> 277 us, prep backward 42 -> 37 us at the benchmark shape.
// Two windows of 8 MB miss the L2: 319 us.)
#include "pclip_common.h"

#define FD 128
#define BWD_WINS 4           // windows per gather launch = partial gradients
#define BWD_SHARE 2          // XCDs on one window (BWD_WINS * BWD_SHARE = the 8 XCDs)
#define BWD_WIN_BYTES ((size_t)4 << 20)

struct FilipBwdArgs {
    const uint16_t* hnP;   // partner images [Np][TPp][128] bf16
    const uint32_t* mwO;   // own mask words [No][TOp/32]
    const uint32_t* mwP;   // partner mask words [Np][TPp/32]
    int No, TOp, Np, TPp;
    const uint16_t* arg1;  // [pair][TOp]  partner token chosen by each own token
    const uint16_t* arg2;  // [pair][TPp]  own token chosen by each partner token
    const uint32_t* mut1;  // [pair][TOp/32] bit a: own token a and the partner token it chose chose each other
    const uint32_t* mut2;  // [pair][TPp/32] the same for the partner tokens
    const float* g1;       // [pair] d loss / d score averaged over OWN tokens
    const float* g2;       // [pair] d loss / d score averaged over PARTNER tokens
    size_t so, sp;         // pair index into arg1 / arg2 = own*so + partner*sp
    size_t gso, gsp;       // pair index into g1 / g2 = own*gso + partner*gsp (partner-contiguous copies)
    const float* t;
    const float* icO;      // [No] 1/max(valid own tokens, 1e-6), 0 where the sequence is empty
    const float* icP;      // [Np] same for partners
    int* seg;              // [No][nwt][TOp+1] bucket offsets inside the window's list
    uint32_t* list;        // [No][Np*TPp] bucketed entries (partner << 16 | partner token); window x of a
                           // sequence starts at entry x*qpw*TPp of its region
    float* dn;             // [nslot][No][TOp][128] f32 out: partial gradients (nslot = min(nwt, BWD_WINS))
    int nwt, qpw;          // partner windows in total (1, or a multiple of BWD_WINS), partners per window
    int win0, accumulate;  // gather launch: first window it covers; add to the partials instead of overwriting
};

// ---- the small jobs in front of the gathers, ONE launch (they are 4-6 us each as launches of
// their own): blocks [0, nC) count tokens, [nC, nC + nT) transpose the upstream gradients,
// the rest add up dt.
struct BwdPreArgs {
    const uint32_t* mwA; const uint32_t* mwB; int nA, ntA, nB, ntB;   // count: inverse valid-token counts
    float* icA; float* icB;
    const float* gA; const float* gB; int R, C;                        // transpose: [R][C] -> [C][R], both matrices
    float* gAT; float* gBT;
    const float* sA; const float* sB; const float* t; float* dt;       // dt += -(1/t) * sum(gA*sA + gB*sB)
    int nC, nT, nD, tiles_x;
};

// inverse valid-token counts of every sequence, both sides
__device__ __forceinline__ void pre_count(const BwdPreArgs& p, int blk) {
    int s = blk * 256 + threadIdx.x;
    const uint32_t* mw = p.mwA;
    float* ic = p.icA;
    int nt = p.ntA;
    if (s >= p.nA) {
        s -= p.nA; mw = p.mwB; ic = p.icB; nt = p.ntB;
        if (s >= p.nB) return;
    }
    int c = 0;
    for (int k = 0; k < nt; ++k) c += __popc(mw[(size_t)s * nt + k]);
    ic[s] = c > 0 ? 1.f / (float)c : 0.f;
}

// [R][C] -> [C][R] for the two upstream-gradient matrices at once (tiny: the B-side pass reads
// them partner-contiguous, as the A-side pass reads the originals)
__device__ __forceinline__ void pre_transpose(const BwdPreArgs& p, int blk, float (&tile)[2][32][33]) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int by = blk / p.tiles_x, bx = blk - by * p.tiles_x;
    const int r0 = by * 32, c0 = bx * 32;
    const int R = p.R, C = p.C;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        const bool in = r < R && c < C;
        tile[0][k][tx] = in ? p.gA[(size_t)r * C + c] : 0.f;
        tile[1][k][tx] = in ? p.gB[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < R && c < C) {
            p.gAT[(size_t)c * R + r] = tile[0][tx][k];
            p.gBT[(size_t)c * R + r] = tile[1][tx][k];
        }
    }
}

// dt += -(1/t) * sum(gA*simA + gB*simB); pairs whose partner is empty hold infinities in the
// reference too and are left out.
__device__ __forceinline__ void pre_dt(const BwdPreArgs& p, int blk, float (&red)[4]) {
    const size_t n = (size_t)p.R * p.C;
    float acc = 0.f;
    for (size_t k = (size_t)blk * 256 + threadIdx.x; k < n; k += (size_t)p.nD * 256) acc += p.gA[k] * p.sA[k] + p.gB[k] * p.sB[k];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(p.dt, -(red[0] + red[1] + red[2] + red[3]) / p.t[0]);
}

__global__ __launch_bounds__(256) void filip_bwd_pre_kernel(BwdPreArgs p) {
    __shared__ float tile[2][32][33];
    __shared__ float red[4];
    const int b = blockIdx.x;
    if (b < p.nC) pre_count(p, b);
    else if (b < p.nC + p.nT) pre_transpose(p, b - p.nC, tile);
    else pre_dt(p, b - p.nC - p.nT, red);
}

// A token pair (a, b) of a sequence pair that chose each other appears twice in the routing,
// with the same partner row: once as a's choice, once in the bucket of the tokens that chose a.
// About half of all tokens of random unit vectors are in such pairs.  This kernel marks them
// (one bit per token and sequence pair, both sides); the inversion then leaves them out of the
// buckets and the gather adds the second weight to the first fetch: a quarter fewer rows.
// One wave per sequence pair; the two arg rows sit in LDS for the cross look-ups.
__global__ __launch_bounds__(256) void filip_mutual_kernel(const uint16_t* __restrict__ argA, const uint16_t* __restrict__ argB,
                                                           const uint32_t* __restrict__ mwA, const uint32_t* __restrict__ mwB, int Br, int Bc,
                                                           int TAp, int TBp, uint32_t* __restrict__ mutA, uint32_t* __restrict__ mutB) {
    extern __shared__ uint16_t sh16[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t pair = (size_t)blockIdx.x * 4 + w;
    if (pair >= (size_t)Br * Bc) return;
    const int i = (int)(pair / Bc), j = (int)(pair - (size_t)i * Bc);
    uint16_t* a_l = sh16 + (size_t)w * (TAp + TBp);
    uint16_t* b_l = a_l + TAp;
    const uint16_t* ga = argA + pair * TAp;
    const uint16_t* gb = argB + pair * TBp;
    const uint32_t* ma = mwA + (size_t)i * (TAp >> 5);
    const uint32_t* mb = mwB + (size_t)j * (TBp >> 5);
    // entries of padding tokens are unspecified: they enter as an index that matches nothing
    // (four entries per lane and load; Tp is a multiple of 32, so the rows are 8-byte aligned)
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
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // 256 tokens at a time, so that the four dependent LDS look-ups of a lane overlap
    auto mark = [&](const uint16_t* own, int TOp, const uint16_t* oth, int TPp, uint32_t* out) {
        for (int base = 0; base < TOp; base += 256) {
            int x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = base + 64 * u + lane;
                x[u] = k < TOp ? (int)own[k] : 0xffff;
            }
            bool m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) m[u] = x[u] < TPp && (int)oth[x[u] < TPp ? x[u] : 0] == base + 64 * u + lane;
            // the eight words of the 256 tokens leave as one store: lane l holds word l
            uint32_t word = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long bal = __ballot(m[u]);
                word = lane == 2 * u ? (uint32_t)bal : lane == 2 * u + 1 ? (uint32_t)(bal >> 32) : word;
            }
            if (lane < 8 && base + 32 * lane < TOp) out[(base >> 5) + lane] = word;
        }
    };
    mark(a_l, TAp, b_l, TBp, mutA + pair * (TAp >> 5));
    mark(b_l, TBp, a_l, TAp, mutB + pair * (TBp >> 5));
}

// One workgroup per (own sequence, partner window): counting sort of the window's valid partner
// tokens by the own token they chose.  Eight u16 entries are read per 16-byte load; the sorted
// entries are built in LDS and leave as one coalesced copy (scattered 4-byte stores to global
// memory made the earlier one-workgroup-per-sequence form four times slower); a window too
// large for the LDS stage scatters straight to global memory instead.
// Dynamic LDS: TOp counters, then `stage` entries.
#define INV_THREADS 1024      // a 4-MB window stages 64 KB: two workgroups per CU, so 1024 threads each for full occupancy
#define INV_STAGE_MAX 24576   // entries (96 KB)
#define INV_KEEP 2            // 8-entry vectors a thread keeps in registers between the two passes
__global__ __launch_bounds__(INV_THREADS) void filip_invert_kernel(FilipBwdArgs p, int stage) {
    extern __shared__ int hist[];                        // [own token] counts, then bucket cursors
    __shared__ int wsum[INV_THREADS / 64];
    uint32_t* sorted = reinterpret_cast<uint32_t*>(hist + p.TOp);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int o = blockIdx.x / p.nwt, x = blockIdx.x - o * p.nwt;
    const int q_lo = x * p.qpw, q_hi = min(p.Np, q_lo + p.qpw);
    const int ntP = p.TPp >> 5;
    const int vpr = p.TPp >> 3;                          // 8-entry vectors per partner row
    const int per = (p.TOp + INV_THREADS - 1) / INV_THREADS;   // consecutive buckets per thread in the scan
    int* seg = p.seg + ((size_t)o * p.nwt + x) * (p.TOp + 1);
    uint32_t* list = p.list + (size_t)o * p.Np * p.TPp + (size_t)q_lo * p.TPp;
    for (int k = tid; k < p.TOp; k += INV_THREADS) hist[k] = 0;
    __syncthreads();
    const bool own_live = p.icO[o] > 0.f;
    const int nvec = (own_live && q_hi > q_lo) ? (q_hi - q_lo) * vpr : 0;
    // the first INV_KEEP vectors of a thread stay in registers for the second pass (a 4-MB window
    // of 256-token partners is exactly two per thread), later ones are read again
    uint4 keep_d[INV_KEEP];
    uint32_t keep_m[INV_KEEP];
#pragma unroll
    for (int u = 0; u < INV_KEEP; ++u) keep_m[u] = 0u;
    for (int v = tid, u = 0; v < nvec; v += INV_THREADS, ++u) {
        const int ql = v / vpr, pt0 = (v - ql * vpr) * 8, q = q_lo + ql;
        const size_t pair = (size_t)o * p.so + (size_t)q * p.sp;
        // valid partner tokens whose choice is not mutual (those ride on the own token's fetch)
        const uint32_t m8 = ((p.mwP[(size_t)q * ntP + (pt0 >> 5)] & ~p.mut2[pair * ntP + (pt0 >> 5)]) >> (pt0 & 31)) & 0xffu;
        if (m8 == 0) continue;
        const uint4 d = *reinterpret_cast<const uint4*>(p.arg2 + pair * p.TPp + pt0);
#pragma unroll
        for (int uu = 0; uu < INV_KEEP; ++uu)
            if (uu == u) { keep_d[uu] = d; keep_m[uu] = m8; }
        const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((m8 >> k) & 1u) atomicAdd(&hist[(dw[k >> 1] >> (16 * (k & 1))) & 0xffffu], 1);
    }
    __syncthreads();
    int total;
    {   // exclusive scan of the counts, `per` consecutive ones per thread
        const int b0 = tid * per;
        int s = 0;
        for (int k = 0; k < per; ++k)
            if (b0 + k < p.TOp) s += hist[b0 + k];
        int incl = s;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
            const int n = __shfl_up(incl, o2);
            if (lane >= o2) incl += n;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int run = incl - s;
        total = 0;
        for (int k = 0; k < INV_THREADS / 64; ++k) {
            if (k < w) run += wsum[k];
            total += wsum[k];
        }
        for (int k = 0; k < per; ++k) {
            const int b = b0 + k;
            if (b >= p.TOp) break;
            const int n = hist[b];
            hist[b] = run;
            seg[b] = run;
            run += n;
        }
        if (tid == 0) seg[p.TOp] = total;
    }
    __syncthreads();
    // (the target is the LDS stage or global memory: written as two stores, since one store through a pointer that may
    // be either is a flat instruction, slower than an LDS write)
    for (int v = tid, u = 0; v < nvec; v += INV_THREADS, ++u) {
        const int ql = v / vpr, pt0 = (v - ql * vpr) * 8, q = q_lo + ql;
        uint32_t m8;
        uint4 d;
        if (u < INV_KEEP) {
            m8 = 0u;
            d = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int uu = 0; uu < INV_KEEP; ++uu)
                if (uu == u) { d = keep_d[uu]; m8 = keep_m[uu]; }
            if (m8 == 0) continue;
        } else {
            const size_t pair = (size_t)o * p.so + (size_t)q * p.sp;
            m8 = ((p.mwP[(size_t)q * ntP + (pt0 >> 5)] & ~p.mut2[pair * ntP + (pt0 >> 5)]) >> (pt0 & 31)) & 0xffu;
            if (m8 == 0) continue;
            d = *reinterpret_cast<const uint4*>(p.arg2 + pair * p.TPp + pt0);
        }
        const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((m8 >> k) & 1u) {
                const int pos = atomicAdd(&hist[(dw[k >> 1] >> (16 * (k & 1))) & 0xffffu], 1);
                const uint32_t ent = ((uint32_t)q << 16) | (uint32_t)(pt0 + k);
                if (stage) sorted[pos] = ent;
                else list[pos] = ent;
            }
    }
    if (stage) {
        __syncthreads();
        for (int k = tid; k < total; k += INV_THREADS) list[k] = sorted[k];
    }
}

// 16 lanes own one gradient row (own sequence o, token a) for ONE partner window: lane l16
// keeps columns 8*l16..+7.  Index/weight loads are spread over the 16 lanes (16 partners or 16
// list entries at a time) and broadcast inside the group; the 16 row gathers of a batch are all
// in flight together.
__device__ __forceinline__ void gather_block(const FilipBwdArgs& p, const int bid) {
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    const int chunks = p.TOp >> 4;
    const int nslot = p.nwt < BWD_WINS ? p.nwt : BWD_WINS;
    const int share = nslot > 1 ? BWD_SHARE : 1;
    // block % 8 is the XCD: its low part picks the window, its high part which chunks of the own sequence
    const int xcd = bid % (nslot * share), blk = bid / (nslot * share);
    const int slot = xcd % nslot, part = xcd / nslot;
    const int x = p.win0 + slot;
    const int cps = (chunks + share - 1) / share;   // chunks per sharing XCD
    const int o = blk / cps;
    const int cidx = (blk - o * cps) * share + part;
    if (cidx >= chunks) return;
    const int a = ((cidx + o) % chunks) * 16 + grp;   // rotated by the sequence: a shader engine sees every chunk position
    const int q_lo = x * p.qpw, q_hi = min(p.Np, q_lo + p.qpw);
    const bool valid = (p.mwO[(size_t)o * (p.TOp >> 5) + (a >> 5)] >> (a & 31)) & 1u;
    // the partials of one gradient row lie next to each other ([row][slot][128]): the pass that
    // sums them reads 4 KB in one piece instead of four rows 33 MB apart
    float4* out = reinterpret_cast<float4*>(p.dn + ((((size_t)o * p.TOp + a) * nslot + slot) * FD) + l16 * 8);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.accumulate && valid) {
        const float4 u = out[0], v = out[1];
        acc[0] = u.x; acc[1] = u.y; acc[2] = u.z; acc[3] = u.w;
        acc[4] = v.x; acc[5] = v.y; acc[6] = v.z; acc[7] = v.w;
    }
    if (valid && q_lo < q_hi) {
        const float inv_t = 1.f / fabsf(p.t[0]);
        const float c1 = p.icO[o] * inv_t;
        // rows are addressed by a scalar base and a 32-bit byte offset per lane (the images of one side stay below
        // 4 GB: checked by the host), computed once per lane and batch and broadcast inside the group
        const unsigned char* const img = reinterpret_cast<const unsigned char*>(p.hnP);
        const uint32_t row_stride = (uint32_t)p.TPp * (FD * 2), lane_off = (uint32_t)l16 * 16u;
        // own-side max terms: one partner row per partner sequence of the window
        for (int q0 = q_lo; q0 < q_hi; q0 += 16) {
            const int q = q0 + l16;
            uint32_t b = (uint32_t)(q_hi - 1) * row_stride;   // filler past the window's end: weight 0, a row inside the window
            float w = 0.f;
            if (q < q_hi) {
                const size_t pair = (size_t)o * p.so + (size_t)q * p.sp;
                b = ((uint32_t)q * (uint32_t)p.TPp + (uint32_t)p.arg1[pair * p.TOp + a]) * (FD * 2);
                const float icq = p.icP[q];
                w = icq > 0.f ? p.g1[(size_t)o * p.gso + (size_t)q * p.gsp] * c1 : 0.f;
                // b chose a as well: its share of the partner-side term comes with this fetch
                if ((p.mut1[pair * (p.TOp >> 5) + (a >> 5)] >> (a & 31)) & 1u) w += p.g2[(size_t)o * p.gso + (size_t)q * p.gsp] * icq * inv_t;
            }
            bf16x8 rows[16];
            float ws[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t bk = (uint32_t)__shfl((int)b, k, 16);
                ws[k] = __shfl(w, k, 16);
                rows[k] = *reinterpret_cast<const bf16x8*>(img + (bk + lane_off));
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += ws[k] * (float)rows[k][e];
        }
        // partner-side max terms of this window that chose this token: its bucket of the sorted list
        const int* seg = p.seg + ((size_t)o * p.nwt + x) * (p.TOp + 1) + a;
        const uint32_t* list = p.list + (size_t)o * p.Np * p.TPp + (size_t)q_lo * p.TPp;
        const int end = seg[1];
        for (int pos0 = seg[0]; pos0 < end; pos0 += 16) {
            const int pos = pos0 + l16;
            uint32_t ent = 0u;   // byte offset of the entry's row in the partner images
            float w = 0.f;
            if (pos < end) {
                const uint32_t e = list[pos];
                const int q = e >> 16;
                ent = ((uint32_t)q * (uint32_t)p.TPp + (e & 0xffffu)) * (FD * 2);
                w = p.g2[(size_t)o * p.gso + (size_t)q * p.gsp] * p.icP[q] * inv_t;
            }
            bf16x8 rows[16];
            float ws[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t ek = __shfl(ent, k, 16);
                ws[k] = __shfl(w, k, 16);
                // a batch is 16 entries; the lanes past the bucket's end fetch nothing (2 % of the kernel)
                if (pos0 + k < end) rows[k] = *reinterpret_cast<const bf16x8*>(img + (ek + lane_off));
                else rows[k] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += ws[k] * (float)rows[k][e];
        }
    }
    if (!valid && p.nwt > 1) return;   // the window sum writes the zero rows of padding tokens itself
    // streaming stores: the partials (268 MB per launch at the benchmark shape) do not push the window's images out of
    // the L2 (gather 346 -> 334 us; the pass that sums them then reads them from HBM, 42 -> 50 us)
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v v0 = {acc[0], acc[1], acc[2], acc[3]}, v1 = {acc[4], acc[5], acc[6], acc[7]};
    __builtin_nontemporal_store(v0, reinterpret_cast<f32x4v*>(out));
    __builtin_nontemporal_store(v1, reinterpret_cast<f32x4v*>(out) + 1);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void filip_gather_kernel(FilipBwdArgs p) {
    gather_block(p, (int)blockIdx.x);
}
// Both sides of the backward in one launch (the first `na` blocks are the A side's; `na` is a multiple of 8, so a
// block's XCD is the same as in a launch of its own): one ramp and one tail instead of two.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void filip_gather2_kernel(FilipBwdArgs pa, FilipBwdArgs pb, int na) {
    if ((int)blockIdx.x < na) gather_block(pa, (int)blockIdx.x);
    else gather_block(pb, (int)blockIdx.x - na);
}

// out = sum of the nslot partial gradients (n4 float4 each); rows of padding tokens (never
// written by the gather) are zero.
__global__ __launch_bounds__(256) void filip_window_sum_kernel(const float4* __restrict__ part, size_t n4, int nslot, float4* __restrict__ out,
                                                               const uint32_t* __restrict__ mwO, int TOp) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n4) return;
    const size_t row = k / (FD / 4);
    const int o = (int)(row / TOp), a = (int)(row - (size_t)o * TOp);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((mwO[(size_t)o * (TOp >> 5) + (a >> 5)] >> (a & 31)) & 1u) {
        // streaming loads: the partials are read once and are larger than the memory-side cache
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const f32x4v* pr = reinterpret_cast<const f32x4v*>(part + row * nslot * (FD / 4) + (k - row * (FD / 4)));   // [row][slot][128]
        f32x4v a = __builtin_nontemporal_load(pr);
        for (int x = 1; x < nslot; ++x) a += __builtin_nontemporal_load(pr + (size_t)x * (FD / 4));
        s = make_float4(a.x, a.y, a.z, a.w);
    }
    out[k] = s;
}

// ---- host side -------------------------------------------------------------------------
static inline int tp32(int T) { return (T + 31) / 32 * 32; }

// Window plan of one gather side.  One window when the partner images fit the budget (4 MB of
// padded images: an XCD's L2), else windows of about that size in multiples of four.
struct BwdPlan { int nwt, qpw; };
static BwdPlan bwd_plan(int Np, int TPp) {
    const size_t per = (size_t)TPp * FD * 2, budget = BWD_WIN_BYTES;
    BwdPlan pl;
    if ((size_t)Np * per <= budget || Np < BWD_WINS) { pl.nwt = 1; pl.qpw = Np; return pl; }
    int qpw = (int)(budget / per);
    if (qpw < 1) qpw = 1;
    int nwt = (Np + qpw - 1) / qpw;
    nwt = (nwt + BWD_WINS - 1) / BWD_WINS * BWD_WINS;
    pl.nwt = nwt; pl.qpw = qpw;
    return pl;
}

struct BwdWs {
    size_t listA, listB, segA, segB, icA, icB, gT, mutA, mutB, partA, partB, total;
    size_t dnA, dnB, total_fused;   // fused entry: scratch dn of a side that runs as a single window
    BwdPlan planA, planB;   // A-side pass (partners = B) and B-side pass (partners = A)
};
static BwdWs bwd_ws_layout(int Br, int TA, int Bc, int TB) {
    const size_t TAp = tp32(TA), TBp = tp32(TB);
    BwdWs w;
    w.planA = bwd_plan(Bc, (int)TBp);
    w.planB = bwd_plan(Br, (int)TAp);
    size_t off = 0;
    w.listA = off; off += pclip_align_up((size_t)Br * Bc * TBp * 4, 256);   // buckets of B tokens per A sequence
    w.listB = off; off += pclip_align_up((size_t)Bc * Br * TAp * 4, 256);   // buckets of A tokens per B sequence
    w.segA = off;  off += pclip_align_up((size_t)Br * w.planA.nwt * (TAp + 1) * 4, 256);
    w.segB = off;  off += pclip_align_up((size_t)Bc * w.planB.nwt * (TBp + 1) * 4, 256);
    w.icA = off;   off += pclip_align_up((size_t)Br * 4, 256);
    w.icB = off;   off += pclip_align_up((size_t)Bc * 4, 256);
    w.gT = off;    off += pclip_align_up((size_t)2 * Br * Bc * 4, 256);   // transposed copies of gA, gB
    w.mutA = off;  off += pclip_align_up((size_t)Br * Bc * (TAp / 32) * 4, 256);   // mutual-choice bits of the A tokens
    w.mutB = off;  off += pclip_align_up((size_t)Br * Bc * (TBp / 32) * 4, 256);
    w.partA = off; off += w.planA.nwt > 1 ? pclip_align_up((size_t)BWD_WINS * Br * TAp * FD * 4, 256) : 0;   // partials of dnA
    w.partB = off; off += w.planB.nwt > 1 ? pclip_align_up((size_t)BWD_WINS * Bc * TBp * FD * 4, 256) : 0;
    w.total = off;
    w.dnA = off; off += w.planA.nwt > 1 ? 0 : pclip_align_up((size_t)Br * TAp * FD * 4, 256);
    w.dnB = off; off += w.planB.nwt > 1 ? 0 : pclip_align_up((size_t)Bc * TBp * FD * 4, 256);
    w.total_fused = off;
    return w;
}

extern "C" size_t pclip_filip_bwd_workspace_bytes(int Br, int TA, int Bc, int TB) {
    if (Br < 1 || Bc < 1 || TA < 1 || TB < 1) return 0;
    return bwd_ws_layout(Br, TA, Bc, TB).total;
}
extern "C" size_t pclip_filip_bwd_fused_workspace_bytes(int Br, int TA, int Bc, int TB) {
    if (Br < 1 || Bc < 1 || TA < 1 || TB < 1) return 0;
    return bwd_ws_layout(Br, TA, Bc, TB).total_fused;
}

// What follows the gathers of one side: nothing but the window sum (dn wanted), or the backward
// of prep straight from the partials (dh wanted; dn is never formed).
struct BwdTail {
    const float* h;      // raw token embeddings [No,T,d], or NULL for "write dn"
    float* dh;
    int T, d, normalise;
    const float* tsign;
};

// blocks of one gather launch of a side
static unsigned bwd_gather_blocks(const FilipBwdArgs& a) {
    const int nslot = a.nwt < BWD_WINS ? a.nwt : BWD_WINS;
    const int share = nslot > 1 ? BWD_SHARE : 1;
    return (unsigned)(a.No * ((a.TOp / 16 + share - 1) / share) * nslot * share);
}
// what follows the gathers of a side: the sum of the partials, or the fused tail
static int bwd_finish_side(const FilipBwdArgs& a, float* part, float* dn, const BwdTail& tail, hipStream_t st) {
    const int nslot = a.nwt < BWD_WINS ? a.nwt : BWD_WINS;
    const size_t n = (size_t)a.No * a.TOp * FD;
    if (tail.h)   // padding rows of the partials were never written: the mask words keep them out
        return filip_prep_bwd_launch(tail.h, a.dn, nslot, FD, (size_t)nslot * FD, a.nwt > 1 ? a.mwO : nullptr, a.No, tail.T, tail.d, tail.normalise,
                                     tail.tsign, 0, tail.dh, st);
    if (a.nwt > 1)
        hipLaunchKernelGGL(filip_window_sum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (const float4*)part, n / 4, nslot,
                           (float4*)dn, a.mwO, a.TOp);
    return PCLIP_OK;
}
// gather launches of one side (four windows each), then the sum of the partials or the fused tail
static int bwd_gather_side(FilipBwdArgs a, float* part, float* dn, const BwdTail& tail, hipStream_t st) {
    a.dn = a.nwt > 1 ? part : dn;
    for (int w0 = 0; w0 < a.nwt; w0 += BWD_WINS) {
        a.win0 = w0;
        a.accumulate = w0 > 0;
        hipLaunchKernelGGL(filip_gather_kernel, dim3(bwd_gather_blocks(a)), dim3(256), 0, st, a);
    }
    return bwd_finish_side(a, part, dn, tail, st);
}

static int filip_bwd_impl(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA, const uint16_t* hnB, const uint32_t* maskwB, int Bc,
                         int TB, const float* t, const float* sim_A, const float* sim_B, const uint16_t* argA, const uint16_t* argB,
                         const float* gA, const float* gB, float* dnA, float* dnB, const BwdTail& tailA, const BwdTail& tailB, float* dt,
                         void* ws, size_t ws_bytes, void* stream, bool marks_in_ws = false) {
    PCLIP_REQUIRE(hnA && maskwA && hnB && maskwB && t && sim_A && sim_B && argA && argB && gA && gB && dt && ws, "filip backward: null pointer");
    PCLIP_REQUIRE(Br >= 1 && Bc >= 1 && Br <= 65535 && Bc <= 65535, "filip backward: Br=%d Bc=%d (need 1..65535)", Br, Bc);
    PCLIP_REQUIRE(TA >= 1 && TA <= PCLIP_FILIP_MAX_T_LONG && TB >= 1 && TB <= PCLIP_FILIP_MAX_T_LONG, "filip backward: TA=%d TB=%d (need 1..%d)",
                  TA, TB, PCLIP_FILIP_MAX_T_LONG);
    // the gather addresses partner rows by a 32-bit byte offset into one side's images
    PCLIP_REQUIRE((size_t)Br * tp32(TA) * FD * 2 <= 0xFFFFFFFFull && (size_t)Bc * tp32(TB) * FD * 2 <= 0xFFFFFFFFull,
                  "filip backward: the token images of one side exceed 4 GB (Br=%d TA=%d Bc=%d TB=%d)", Br, TA, Bc, TB);
    const BwdWs L = bwd_ws_layout(Br, TA, Bc, TB);
    const size_t need = (tailA.h || tailB.h) ? L.total_fused : L.total;
    PCLIP_REQUIRE(ws_bytes >= need, "filip backward: workspace %zu < %zu bytes", ws_bytes, need);
    if (tailA.h) dnA = (float*)((char*)ws + L.dnA);   // scratch, used only when the side runs as one window
    if (tailB.h) dnB = (float*)((char*)ws + L.dnB);
    const int TAp = tp32(TA), TBp = tp32(TB);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* icA = (float*)(w + L.icA);
    float* icB = (float*)(w + L.icB);
    float* gAT = (float*)(w + L.gT);
    float* gBT = gAT + (size_t)Br * Bc;
    {
        BwdPreArgs q;
        q.mwA = maskwA; q.mwB = maskwB; q.nA = Br; q.ntA = TAp / 32; q.nB = Bc; q.ntB = TBp / 32; q.icA = icA; q.icB = icB;
        q.gA = gA; q.gB = gB; q.R = Br; q.C = Bc; q.gAT = gAT; q.gBT = gBT;
        q.sA = sim_A; q.sB = sim_B; q.t = t; q.dt = dt;
        q.nC = (Br + Bc + 255) / 256;
        q.tiles_x = (Bc + 31) / 32;
        q.nT = dnB ? q.tiles_x * ((Br + 31) / 32) : 0;            // the transposed copies serve the B side
        const size_t n = (size_t)Br * Bc;
        q.nD = dnA ? (int)((n + 255) / 256 > 256 ? 256 : (n + 255) / 256) : 0;   // dt goes with the A side
        hipLaunchKernelGGL(filip_bwd_pre_kernel, dim3(q.nC + q.nT + q.nD), dim3(256), 0, st, q);
        if (int rc = pclip_check_launch("filip_bwd_pre_kernel")) return rc;
    }
    uint32_t* mutA = (uint32_t*)(w + L.mutA);
    uint32_t* mutB = (uint32_t*)(w + L.mutB);
    if (!marks_in_ws) {
        const size_t lds = (size_t)4 * (TAp + TBp) * sizeof(uint16_t);
        static PclipLdsLimit lds_limit;
        if (int rc = pclip_ensure_lds(lds_limit, (const void*)filip_mutual_kernel, lds, "filip_mutual_kernel")) return rc;
        const size_t npair = (size_t)Br * Bc;
        hipLaunchKernelGGL(filip_mutual_kernel, dim3((unsigned)((npair + 3) / 4)), dim3(256), lds, st, argA, argB, maskwA, maskwB, Br, Bc, TAp, TBp,
                           mutA, mutB);
        if (int rc = pclip_check_launch("filip_mutual_kernel")) return rc;
    }
    FilipBwdArgs a, b;
    // gradient of the A images: own = A(i), partner = B(j), pair = i*Bc + j
    a.hnP = hnB; a.mwO = maskwA; a.mwP = maskwB; a.No = Br; a.TOp = TAp; a.Np = Bc; a.TPp = TBp;
    a.arg1 = argA; a.arg2 = argB; a.mut1 = mutA; a.mut2 = mutB; a.g1 = gA; a.g2 = gB; a.so = (size_t)Bc; a.sp = 1; a.gso = (size_t)Bc; a.gsp = 1; a.t = t;
    a.icO = icA; a.icP = icB; a.seg = (int*)(w + L.segA); a.list = (uint32_t*)(w + L.listA); a.dn = nullptr;
    a.nwt = L.planA.nwt; a.qpw = L.planA.qpw; a.win0 = 0; a.accumulate = 0;
    // gradient of the B images: own = B(j), partner = A(i), pair = i*Bc + j
    b.hnP = hnA; b.mwO = maskwB; b.mwP = maskwA; b.No = Bc; b.TOp = TBp; b.Np = Br; b.TPp = TAp;
    b.arg1 = argB; b.arg2 = argA; b.mut1 = mutB; b.mut2 = mutA; b.g1 = gBT; b.g2 = gAT; b.so = 1; b.sp = (size_t)Bc; b.gso = (size_t)Br; b.gsp = 1; b.t = t;
    b.icO = icB; b.icP = icA; b.seg = (int*)(w + L.segB); b.list = (uint32_t*)(w + L.listB); b.dn = nullptr;
    b.nwt = L.planB.nwt; b.qpw = L.planB.qpw; b.win0 = 0; b.accumulate = 0;
    auto invert = [&](const FilipBwdArgs& u) -> int {
        const size_t cap = (size_t)u.qpw * u.TPp;                       // entries of one window at most
        const int stage = cap <= INV_STAGE_MAX ? (int)cap : 0;
        const size_t lds = ((size_t)u.TOp + stage) * 4;
        static PclipLdsLimit lds_limit;
        if (int rc = pclip_ensure_lds(lds_limit, (const void*)filip_invert_kernel, lds, "filip_invert_kernel")) return rc;
        hipLaunchKernelGGL(filip_invert_kernel, dim3((unsigned)((size_t)u.No * u.nwt)), dim3(INV_THREADS), lds, st, u, stage);
        return PCLIP_OK;
    };
    if (dnA && dnB && a.nwt <= BWD_WINS && b.nwt <= BWD_WINS && bwd_gather_blocks(a) % 8 == 0) {
        // both sides wanted and one gather launch each: the two gathers share a launch
        if (int rc = invert(a)) return rc;
        if (int rc = invert(b)) return rc;
        a.dn = a.nwt > 1 ? (float*)(w + L.partA) : dnA;
        b.dn = b.nwt > 1 ? (float*)(w + L.partB) : dnB;
        const unsigned na = bwd_gather_blocks(a), nb = bwd_gather_blocks(b);
        hipLaunchKernelGGL(filip_gather2_kernel, dim3(na + nb), dim3(256), 0, st, a, b, (int)na);
        if (int rc = bwd_finish_side(a, (float*)(w + L.partA), dnA, tailA, st)) return rc;
        if (int rc = bwd_finish_side(b, (float*)(w + L.partB), dnB, tailB, st)) return rc;
        return pclip_check_launch("filip backward (both sides)");
    }
    for (int side = 0; side < 2; ++side) {
        if (!(side == 0 ? dnA : dnB)) continue;   // pclip_filip_sim_bwd: that side was not asked for
        const FilipBwdArgs& u = side == 0 ? a : b;
        if (int rc = invert(u)) return rc;
        if (int rc = bwd_gather_side(u, (float*)(w + (side == 0 ? L.partA : L.partB)), side == 0 ? dnA : dnB, side == 0 ? tailA : tailB, st))
            return rc;
        if (int rc = pclip_check_launch(side == 0 ? "filip backward (A side)" : "filip backward (B side)")) return rc;
    }
    return PCLIP_OK;
}

extern "C" int pclip_filip_sim_bwd(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA, const uint16_t* hnB, const uint32_t* maskwB,
                                   int Bc, int TB, const float* t, const float* sim_A, const float* sim_B, const uint16_t* argA,
                                   const uint16_t* argB, const float* gA, const float* gB, float* dnA, float* dnB, float* dt, void* ws,
                                   size_t ws_bytes, void* stream) {
    PCLIP_REQUIRE(dnA || dnB, "filip_sim_bwd: null pointer (dnA and dnB)");
    const BwdTail none = {nullptr, nullptr, 0, 0, 0, nullptr};
    return filip_bwd_impl(hnA, maskwA, Br, TA, hnB, maskwB, Bc, TB, t, sim_A, sim_B, argA, argB, gA, gB, dnA, dnB, none, none, dt, ws, ws_bytes,
                          stream);
}

extern "C" int pclip_filip_sim_bwd_resume(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA, const uint16_t* hnB,
                                          const uint32_t* maskwB, int Bc, int TB, const float* t, const float* sim_A, const float* sim_B,
                                          const uint16_t* argA, const uint16_t* argB, const float* gA, const float* gB, float* dnA,
                                          float* dnB, float* dt, void* ws, size_t ws_bytes, void* stream) {
    PCLIP_REQUIRE(dnA || dnB, "filip_sim_bwd_resume: null pointer (dnA and dnB)");
    const BwdTail none = {nullptr, nullptr, 0, 0, 0, nullptr};
    return filip_bwd_impl(hnA, maskwA, Br, TA, hnB, maskwB, Bc, TB, t, sim_A, sim_B, argA, argB, gA, gB, dnA, dnB, none, none, dt, ws, ws_bytes,
                          stream, true);
}

extern "C" int pclip_filip_bwd_fused(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA, const uint16_t* hnB, const uint32_t* maskwB,
                                     int Bc, int TB, const float* t, const float* sim_A, const float* sim_B, const uint16_t* argA,
                                     const uint16_t* argB, const float* gA, const float* gB, const float* hA, const float* hB, int d,
                                     int normalise, float* dhA, float* dhB, float* dt, void* ws, size_t ws_bytes, void* stream) {
    PCLIP_REQUIRE(hA && hB && dhA && dhB, "filip_bwd_fused: null pointer");
    PCLIP_REQUIRE(d >= 1 && d <= PCLIP_FILIP_MAX_D, "filip_bwd_fused: d=%d (need 1 <= d <= %d)", d, PCLIP_FILIP_MAX_D);
    const BwdTail tA = {hA, dhA, TA, d, normalise, t}, tB = {hB, dhB, TB, d, normalise, nullptr};
    return filip_bwd_impl(hnA, maskwA, Br, TA, hnB, maskwB, Bc, TB, t, sim_A, sim_B, argA, argB, gA, gB, nullptr, nullptr, tA, tB, dt, ws,
                          ws_bytes, stream);
}
