// The whole one-GPU CLIP step in ONE launch (B <= 512, d <= 128), gfx950:
// normalise both towers (modules/models.py:50-51), symmetric InfoNCE (modules/training_utils.py:
// 85-89) and its backward down to the raw head outputs and the temperature.
//
// At B = 256 the step is 0.5 MB of data and 50 MFLOP: latency, not throughput.  As separate
// launches it was five dependent kernels of 4-19 us; here one persistent grid of (B/32)^2
// workgroups (one 32 x 32 logit tile each, all resident at once) runs three phases separated by
// two grid-wide barriers:
//   A  raw rows -> normalised rows in LDS -> the tile T[j][i] = y_j . x_i on exact-f32 matrix cores
//      (v_mfma_f32_32x32x2_f32, the contraction split four ways over the waves and summed through
//      LDS) -> per-tile soft-max statistics of BOTH directions (rows over registers, columns
//      across lanes) from the one tile.  The tile stays in registers.
//   B  log-sum-exp of the tile's rows and columns from everybody's statistics -> G = dLoss/dT ->
//      dX^T += Y^T G^T and dY += G X on the matrix cores (the first takes G straight from the
//      accumulator registers, the second through a transposed copy in LDS) -> partial gradient
//      slabs, one per tile column / row.
//   C  slabs summed through the normalise Jacobian into d e_pep, d e_rec, dt; the loss.
// Data written by one workgroup and read by another inside the launch (tile statistics, the
// diagonal, log-sum-exps, gradient slabs) is stored write-through (sc1) and loaded past the L1
// (sc1), every storing wave drains its stores before the workgroup's one arrival on the barrier
// counter: no cache-wide release / acquire fences (several us per barrier here), and no
// assumption about where a workgroup runs.
//
// The barrier words live in a small device array of this code object (one copy per GPU, zero when
// the code object is loaded): [0] arrivals, [1] a sticky flag set when a wait gave up, [2]
// departures.  The workgroup that departs last clears [0] and [2], so a launch needs no memset
// in front of it.  A grid barrier only completes when the whole grid is resident; launches of
// this kernel from different streams of one process are therefore chained on the host (an
// event from the stream used last), and a launch that still cannot complete (another process
// filling the GPU with a barrier kernel of its own) gives up after about half a second, sets
// the sticky flag and writes NaN to the loss and every gradient — as does every later launch
// on that GPU until pclip_clip_step_sync_reset().
#include "pclip_common.h"

#define CS_D 128            // padded contraction width
#define CS_LD 129           // row pitch of the LDS tiles (floats): odd, so a column read is conflict-free
#define CS_MAXT 16          // tiles per side at most (B <= 512)

struct ClipStepArgs {
    const float* e_pep; const float* e_rec; const float* t;
    float* loss; float* de_pep; float* de_rec; float* dt;
    int B, d, nT;
    int raw;                  // 1: e_pep / e_rec are raw head outputs (normalise, scale, Jacobian, dt); 0: they are x, y as the towers return them
    float scale;              // 0.5 / B
    unsigned* counter;        // [0] arrivals, [1] sticky: a wait gave up, [2] departures (the last one clears [0] and [2])
    unsigned* stamps;         // diagnostic build only: the head of the caller's workspace
    unsigned extra;           // added to the arrivals a barrier waits for (0; 1 in the self-test of the give-up path)
    float2* part_row;         // [nT (tile column)][B]  (max, sum) of row i over the tile's 32 columns
    float2* part_col;         // [nT (tile row)][B]     (max, sum) of column j over the tile's 32 rows
    float* diag;              // [B] S_ii
    float* lse_row; float* lse_col;   // [B]
    float* slab_x;            // [nT (tile column)][B][d]  partial d x
    float* slab_y;            // [nT (tile row)][B][d]     partial d y
};

// Cross-workgroup data: write-through stores, L1-bypassing loads (agent scope, relaxed).
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ void st_f32(float* p, float v) { __hip_atomic_store((gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_f32(const float* p) { return __uint_as_float(__hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void st_f32x2(float* p, float a, float b) {
    __hip_atomic_store((gu64*)p, ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_f32x2(const float* p) {
    const unsigned long long v = __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

// All workgroups of the grid (resident together: the launcher keeps the grid within the CU
// count) meet here.  What a workgroup stored with the helpers above before the barrier, any
// workgroup reads with them after it.  `target` = workgroups x barriers passed so far.
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* gave_up_lds, unsigned long long* stamp = nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x == 0) {
        if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
        const unsigned before = __hip_atomic_fetch_add((gu32*)counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stamp) { stamp[1] = __builtin_amdgcn_s_memrealtime() + (before & 0); stamp[2] = before; }
        unsigned spins = 0;
        for (;;) {   // arrivals and the sticky flag are one 8-byte word: the flag costs no load of its own
            const unsigned long long v = __hip_atomic_load((gu64*)counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) != 0u) { *gave_up_lds = 1; break; }   // a wait gave up, in this launch or an earlier one
            if ((unsigned)v >= target) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 23)) {   // never in a healthy run (about half a second): leave a sticky mark instead of hanging
                __hip_atomic_store((gu32*)(counter + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *gave_up_lds = 1;
                break;
            }
        }
    }
    __syncthreads();
    return *gave_up_lds != 0;
}

// Sum over each group of 16 consecutive lanes (every lane of a group gets it).
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// One wave: 8 of the tile's 32 rows of both towers -> normalised, scaled rows in LDS.  A row is
// held by 16 lanes (columns 4 l .. 4 l + 3 and 64 + 4 l .. 64 + 4 l + 3 of lane l), four rows per
// pass: four passes, all eight loads issued before the first reduction (one memory latency).
__device__ __forceinline__ void stage_rows(const float* __restrict__ e0, int r0, float* tile0, const float* __restrict__ e1, int r1, float* tile1,
                                           int B, int d, float s, bool raw, int w, int lane) {
    const int l = lane & 15, g = lane >> 4;
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // u: tower (u >> 2), pass ((u >> 1) & 1), column half (u & 1)
        const int r = 8 * w + 4 * ((u >> 1) & 1) + g;
        const int row = ((u >> 2) ? r1 : r0) + r, col = 64 * (u & 1) + 4 * l;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < B && col < d) v[u] = *reinterpret_cast<const float4*>(((u >> 2) ? e1 : e0) + (size_t)row * d + col);
    }
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
        const float4 a = v[u], b = v[u + 1];
        const float ss = sum16(((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w)));
        const float f = raw ? s / fmaxf(sqrtf(ss), 1e-12f) : 1.f;
        float* o = ((u >> 2) ? tile1 : tile0) + (8 * w + 4 * ((u >> 1) & 1) + g) * CS_LD + 4 * l;
        o[0] = a.x * f; o[1] = a.y * f; o[2] = a.z * f; o[3] = a.w * f;
        o[64] = b.x * f; o[65] = b.y * f; o[66] = b.z * f; o[67] = b.w * f;
    }
}

#ifdef CLIP_STAMPS   // diagnostic build: 100-MHz time stamps of every workgroup, 16 per workgroup, after the counter block
#define CS_COUNTER_BYTES (256 + 256 * 16 * 8)
#define CSTAMP_AT(k) (reinterpret_cast<unsigned long long*>(p.stamps) + 32 + blockIdx.x * 16 + (k))
#define CSTAMP(k) do { if (threadIdx.x == 0) *CSTAMP_AT(k) = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CS_COUNTER_BYTES 256
#define CSTAMP_AT(k) nullptr
#define CSTAMP(k) do {} while (0)
#endif

// The visible failure: NaN in the loss, in every gradient row this workgroup would have written, and in dt.
__device__ __forceinline__ void clip_step_poison(const ClipStepArgs& p, int tid, unsigned nwg) {
    const float nan = __int_as_float(0x7FC00000);
    const int lane = tid & 63, w = tid >> 6, l = lane & 15, g = lane >> 4;
    for (int row2 = (4 * w + g) * (int)nwg + (int)blockIdx.x; row2 < 2 * p.B; row2 += 16 * (int)nwg) {
        const bool rec = row2 >= p.B;
        float* der = (rec ? p.de_rec : p.de_pep) + (size_t)(rec ? row2 - p.B : row2) * p.d;
        for (int col = l; col < p.d; col += 16) der[col] = nan;
    }
    if (blockIdx.x == 0 && tid == 0) {
        p.loss[0] = nan;
        if (p.raw) p.dt[0] = nan;
    }
}

__global__ __launch_bounds__(256) void clip_step_kernel(ClipStepArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                       // [32][CS_LD]  normalised pep rows of tile column `it`
    float* ys = xs + 32 * CS_LD;           // [32][CS_LD]  normalised rec rows of tile row `jt`
    float* red = ys + 32 * CS_LD;          // [4][16][64] partial tiles, later [32][CS_LD] output staging
    float* gT = red + 32 * CS_LD;          // [32][33]     G[j][i]
    float* lseR = gT + 32 * 33;            // [32]
    float* lseC = lseR + 32;               // [32]
    int* gave_up = reinterpret_cast<int*>(lseC + 32);   // [1] a barrier wait of this workgroup gave up
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int it = blockIdx.x % p.nT, jt = blockIdx.x / p.nT;
    const int i0 = it * 32, j0 = jt * 32;
    const int B = p.B, d = p.d;
    const unsigned nwg = gridDim.x;

    if (blockIdx.x == 0 && tid == 0 && p.raw) st_f32(p.dt, 0.f);   // phase C adds into it, two barriers from here
    // (a wait that gave up on this GPU earlier left a sticky flag: the barrier words are then in no known state, and the
    // first barrier of every later launch sees the flag and fails visibly)
    if (tid == 0) *gave_up = 0;

    // ---- phase A --------------------------------------------------------------------------
    CSTAMP(0);
    const bool raw = p.raw != 0;
    const float s = raw ? expf(p.t[0] * 0.5f) : 1.f;
    stage_rows(p.e_pep, i0, xs, p.e_rec, j0, ys, B, d, s, raw, w, lane);
    __syncthreads();
    CSTAMP(1);
    f32x16 T = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    {   // this wave's quarter of the contraction: k = 32 w .. 32 w + 31, two per instruction
        const float* xa = xs + c * CS_LD + 32 * w + h;
        const float* ya = ys + c * CS_LD + 32 * w + h;
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) T = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[2 * sidx], xa[2 * sidx], T, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(w * 16 + q) * 64 + lane] = T[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) T[q] = (red[q * 64 + lane] + red[(16 + q) * 64 + lane]) + (red[(32 + q) * 64 + lane] + red[(48 + q) * 64 + lane]);
    // T[q] = S[i0 + c][j0 + acc_row(q, h)], the same in all four waves
    CSTAMP(2);
    const bool ivalid = i0 + c < B;
    if (w == 0) {          // rows of S: over the 16 registers, then the other lane half
        float m = -FLT_MAX;
#pragma unroll
        for (int q = 0; q < 16; ++q) m = fmaxf(m, (j0 + acc_row(q, h) < B) ? T[q] : -FLT_MAX);
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += (j0 + acc_row(q, h) < B) ? expf(T[q] - m) : 0.f;
        sum += __shfl_xor(sum, 32);
        if (h == 0 && ivalid) st_f32x2(reinterpret_cast<float*>(p.part_row + (size_t)jt * B + i0 + c), m, sum);
    } else if (w & 1) {    // columns of S: register q of lane half h is column j0 + acc_row(q, h), over the 32 lanes;
                           // wave 1 takes registers 0-7, wave 3 registers 8-15
        const int rr = (c & 3) + 4 * (c >> 3);   // acc_row(rr, (c >> 2) & 1) == c
        float mc = -FLT_MAX, ec = 0.f;
        const int qb = 4 * (w - 1);
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            const int q = qb + qq;
            const float v = ivalid ? (w == 1 ? T[qq] : T[8 + qq]) : -FLT_MAX;
            float m = v;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
            float e = ivalid ? expf(v - m) : 0.f;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) e += __shfl_xor(e, o);
            mc = (q == rr) ? m : mc;
            ec = (q == rr) ? e : ec;
        }
        // one store instruction for the 32 columns (whole cache lines), from the lane half that holds column c
        if (((c >> 2) & 1) == h && (rr >> 3) == (w >> 1) && j0 + c < B) st_f32x2(reinterpret_cast<float*>(p.part_col + (size_t)it * B + j0 + c), mc, ec);
    }
    if (w == 2 && it == jt) {   // the diagonal S_ii: in the lane half h with acc_row(q, h) == c
        const int rr = (c & 3) + 4 * (c >> 3);
        float dv = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) dv = (q == rr) ? T[q] : dv;
        if (((c >> 2) & 1) == h && ivalid) st_f32(p.diag + i0 + c, dv);
    }
    CSTAMP(3);
    // (a wait that gives up here is noticed behind the second barrier: phase B then runs on garbage, which stays inside the scratch)
    grid_barrier(p.counter, nwg + p.extra, gave_up, CSTAMP_AT(10));
    CSTAMP(4);

    // ---- phase B --------------------------------------------------------------------------
    if (tid < 64) {   // log-sum-exp of this tile's 32 rows (threads 0-31) and 32 columns (32-63)
        const bool col = tid >= 32;
        const int idx = (col ? j0 : i0) + (tid & 31);
        float lse = 0.f;
        if (idx < B) {
            const float2* part = (col ? p.part_col : p.part_row) + idx;
            float2 v[CS_MAXT];   // every tile's statistics requested at once
#pragma unroll
            for (int k = 0; k < CS_MAXT; ++k) v[k] = k < p.nT ? ld_f32x2(reinterpret_cast<const float*>(part + (size_t)k * B)) : make_float2(-FLT_MAX, 0.f);
            float M = -FLT_MAX;
#pragma unroll
            for (int k = 0; k < CS_MAXT; ++k) M = fmaxf(M, v[k].x);
            float S = 0.f;
#pragma unroll
            for (int k = 0; k < CS_MAXT; ++k) S += v[k].y * expf(v[k].x - M);
            lse = M + logf(S);
            // one workgroup per row block / column block records it for the loss
            if (!col && jt == 0) st_f32(p.lse_row + idx, lse);
            if (col && it == 0) st_f32(p.lse_col + idx, lse);
        }
        (col ? lseC : lseR)[tid & 31] = lse;
    }
    __syncthreads();
    CSTAMP(5);
    {
        const float lr = lseR[c];
        const int tdiag = i0 + c - j0;   // local column holding the target of this lane's row
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int jl = acc_row(q, h);
            float g = p.scale * (expf(T[q] - lr) + expf(T[q] - lseC[jl]));
            if (jl == tdiag) g -= 2.f * p.scale;
            T[q] = (ivalid && j0 + jl < B) ? g : 0.f;
        }
    }
    // G[j][i] for the second product
    if (w == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) gT[acc_row(q, h) * 33 + c] = T[q];
    }
    // d x^T [column 32 w + m][i] = sum_j y[j][32 w + m] G[j][i]: G's registers are the B operand as they stand
    f32x16 dxa = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 16; ++q) dxa = __builtin_amdgcn_mfma_f32_32x32x2f32(ys[acc_row(q, h) * CS_LD + 32 * w + c], T[q], dxa, 0, 0, 0);
    __syncthreads();   // gT complete; red (the partial tiles) no longer read
    // d y [j][column 32 w + n] = sum_i G[j][i] x[i][32 w + n]
    f32x16 dya = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int sidx = 0; sidx < 16; ++sidx)
        dya = __builtin_amdgcn_mfma_f32_32x32x2f32(gT[c * 33 + 2 * sidx + h], xs[(2 * sidx + h) * CS_LD + 32 * w + c], dya, 0, 0, 0);
    // Both products leave through LDS as whole rows, stored write-through in 8-byte pieces.
    // d x: register q of lane (c, h) is (row i0 + c, column 32 w + acc_row(q, h));
    // d y: register q of lane (c, h) is (row j0 + acc_row(q, h), column 32 w + c).
    {
        float* st = red;   // [32 rows][CS_LD]
#pragma unroll
        for (int q = 0; q < 16; ++q) st[c * CS_LD + 32 * w + acc_row(q, h)] = dxa[q];
        __syncthreads();
        // wave w: rows w, w + 4, ...; lane: one pair of columns (a whole row per store instruction)
        float* out = p.slab_x + ((size_t)jt * B + i0) * d + 2 * lane;
        if (2 * lane < d) {
#pragma unroll
            for (int r = w; r < 32; r += 4)
                if (i0 + r < B) st_f32x2(out + (size_t)r * d, st[r * CS_LD + 2 * lane], st[r * CS_LD + 2 * lane + 1]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) st[acc_row(q, h) * CS_LD + 32 * w + c] = dya[q];
        __syncthreads();
        out = p.slab_y + ((size_t)it * B + j0) * d + 2 * lane;
        if (2 * lane < d) {
#pragma unroll
            for (int r = w; r < 32; r += 4)
                if (j0 + r < B) st_f32x2(out + (size_t)r * d, st[r * CS_LD + 2 * lane], st[r * CS_LD + 2 * lane + 1]);
        }
    }
    CSTAMP(6);
    const bool poisoned = grid_barrier(p.counter, 2 * nwg + p.extra, gave_up, CSTAMP_AT(13));   // the flag set at either barrier
    CSTAMP(7);
    if (poisoned) {   // uniform over the workgroup
        clip_step_poison(p, tid, nwg);
        return;
    }
    // Every workgroup has left both waits once `nwg` departures are counted: the last one clears the words for the
    // next launch.  The atomic is issued here and its result used at the end of the kernel.
    unsigned departed = 0;
    if (tid == 0) departed = __hip_atomic_fetch_add((gu32*)(p.counter + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- phase C --------------------------------------------------------------------------
    // sixteen lanes per row of the two towers (columns 4 l .. 4 l + 3 and 64 + 4 l .. of lane l), sixteen
    // rows per workgroup and pass: sum of the slabs through the normalise Jacobian
    float dts = 0.f;
    {
        const int l = lane & 15, g = lane >> 4;
        for (int row2 = (4 * w + g) * (int)nwg + (int)blockIdx.x; row2 < 2 * B; row2 += 16 * (int)nwg) {
            const bool rec = row2 >= B;
            const int row = rec ? row2 - B : row2;
            const float* er = (rec ? p.e_rec : p.e_pep) + (size_t)row * d + 4 * l;
            const float* dr = (rec ? p.slab_y : p.slab_x) + (size_t)row * d + 4 * l;
            float* der = (rec ? p.de_rec : p.de_pep) + (size_t)row * d + 4 * l;
            const bool lo = 4 * l < d, hi = 64 + 4 * l < d;
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, a0 = v0, a1 = v0;
            if (lo) v0 = *reinterpret_cast<const float4*>(er);
            if (hi) v1 = *reinterpret_cast<const float4*>(er + 64);
            float2 a[CS_MAXT][4];   // every slab's row requested at once, summed in slab order
#pragma unroll
            for (int k = 0; k < CS_MAXT; ++k) {
                const float* dk = dr + (size_t)k * B * d;
                const bool on = k < p.nT;
                a[k][0] = on && lo ? ld_f32x2(dk) : make_float2(0.f, 0.f);
                a[k][1] = on && lo ? ld_f32x2(dk + 2) : make_float2(0.f, 0.f);
                a[k][2] = on && hi ? ld_f32x2(dk + 64) : make_float2(0.f, 0.f);
                a[k][3] = on && hi ? ld_f32x2(dk + 66) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < CS_MAXT; ++k) {
                a0.x += a[k][0].x; a0.y += a[k][0].y; a0.z += a[k][1].x; a0.w += a[k][1].y;
                a1.x += a[k][2].x; a1.y += a[k][2].y; a1.z += a[k][3].x; a1.w += a[k][3].y;
            }
            const float ss = sum16(((v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w)) + ((v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w)));
            const float dot = sum16(((v0.x * a0.x + v0.y * a0.y) + (v0.z * a0.z + v0.w * a0.w)) + ((v1.x * a1.x + v1.y * a1.y) + (v1.z * a1.z + v1.w * a1.w)));   // e . dx
            const float nrm = sqrtf(ss);
            float f, gcoef;   // d e = f * (dx - gcoef * e)
            if (!raw) {   // the inputs are the normalised rows themselves
                f = 1.f; gcoef = 0.f;
            } else if (nrm >= 1e-12f) {
                const float inv = 1.f / nrm, ndx = dot * inv;   // n . dx
                f = s * inv; gcoef = inv * ndx;
                if (l == 0) dts += 0.5f * s * ndx;
            } else {   // norm clamped: x = e * (s / 1e-12), a plain scale
                f = s / 1e-12f; gcoef = 0.f;
                if (l == 0) dts += 0.5f * f * dot;
            }
            if (lo) *reinterpret_cast<float4*>(der) = make_float4((a0.x - v0.x * gcoef) * f, (a0.y - v0.y * gcoef) * f, (a0.z - v0.z * gcoef) * f, (a0.w - v0.w * gcoef) * f);
            if (hi) *reinterpret_cast<float4*>(der + 64) = make_float4((a1.x - v1.x * gcoef) * f, (a1.y - v1.y * gcoef) * f, (a1.z - v1.z * gcoef) * f, (a1.w - v1.w * gcoef) * f);
        }
        dts = wave_sum(dts);
    }
    {
        __syncthreads();
        float* dtp = lseR;   // free by now
        if (lane == 0) dtp[w] = dts;
        __syncthreads();
        if (tid == 0) {
            const float v = (dtp[0] + dtp[1]) + (dtp[2] + dtp[3]);
            if (raw && v != 0.f) atomicAdd(p.dt, v);
        }
    }
    CSTAMP(8);
    if (blockIdx.x == 0) {   // the loss, summed in a fixed order
        float acc = 0.f;
        for (int i = tid; i < B; i += 256) acc += ld_f32(p.lse_row + i) + ld_f32(p.lse_col + i) - 2.f * ld_f32(p.diag + i);
        acc = wave_sum(acc);
        __syncthreads();
        if (lane == 0) lseC[w] = acc;
        __syncthreads();
        if (tid == 0) p.loss[0] = ((lseC[0] + lseC[1]) + (lseC[2] + lseC[3])) * p.scale;
    }
    CSTAMP(9);
    if (tid == 0 && departed == nwg - 1) {
        __hip_atomic_store((gu32*)p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((gu32*)(p.counter + 2), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- host side -------------------------------------------------------------------------
#define CS_LDS_BYTES ((size_t)(3 * 32 * CS_LD + 32 * 33 + 64 + 4) * sizeof(float))

// The barrier words of the current device's copy of this code object.
static __device__ __align__(256) unsigned clip_sync_words[64];

// Per device: the workgroups of this kernel that the device keeps resident at once (the grid barrier needs the whole
// grid resident: a device partition with few CUs takes the separate launches instead), the device address of the
// barrier words, and the stream the kernel was launched on last.  Asked once per device.
struct ClipStepDevice {
    int cap = 0;                    // 0: not asked yet, -1: the one-launch form is unavailable
    unsigned* sync = nullptr;
    bool launched = false;          // a launch outside stream capture has happened; `last` is its stream
    hipStream_t last = nullptr;
    hipEvent_t chain = nullptr;     // made with the rest of this record, so that no launch ever creates anything
};
static std::mutex g_cs_mu;
static ClipStepDevice g_cs_dev[PCLIP_MAX_DEVICES];
static int g_cs_selftest_extra = 0;   // pclip_clip_step_selftest_giveup

static ClipStepDevice* clip_step_device_locked() {   // g_cs_mu held
    static PclipLdsLimit lim;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PCLIP_MAX_DEVICES) return nullptr;
    ClipStepDevice& D = g_cs_dev[dev];
    if (D.cap == 0) {
        int cus = 0, per_cu = 0;
        void* sp = nullptr;
        D.cap = -1;
        if (pclip_ensure_lds(lim, (const void*)clip_step_kernel, CS_LDS_BYTES, "clip_step_kernel") == PCLIP_OK &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)clip_step_kernel, 256, CS_LDS_BYTES) == hipSuccess && cus * per_cu > 0 &&
            hipGetSymbolAddress(&sp, HIP_SYMBOL(clip_sync_words)) == hipSuccess && sp &&
            hipEventCreateWithFlags(&D.chain, hipEventDisableTiming) == hipSuccess) {
            D.cap = cus * per_cu;
            D.sync = (unsigned*)sp;
        }
        (void)hipGetLastError();
    }
    return &D;
}
static int clip_step_resident_capacity() {
    std::lock_guard<std::mutex> lock(g_cs_mu);
    const ClipStepDevice* D = clip_step_device_locked();
    return D && D->cap > 0 ? D->cap : 0;
}

bool clip_step_fused_ok(int B, int d) {
    if (!(B >= 1 && B <= 32 * CS_MAXT && d >= 4 && d <= CS_D && d % 4 == 0)) return false;
    const int nT = (B + 31) / 32;
    return nT * nT <= clip_step_resident_capacity();
}

struct ClipStepWs { size_t counter, part_row, part_col, diag, lse_row, lse_col, slab_x, slab_y, total; };
static ClipStepWs clip_step_layout(int B, int d) {
    const size_t nT = (B + 31) / 32;
    ClipStepWs L;
    size_t off = 0;
    L.counter = off; off += CS_COUNTER_BYTES;
    L.part_row = off; off += pclip_align_up(nT * B * sizeof(float2), 256);
    L.part_col = off; off += pclip_align_up(nT * B * sizeof(float2), 256);
    L.diag = off; off += pclip_align_up((size_t)B * 4, 256);
    L.lse_row = off; off += pclip_align_up((size_t)B * 4, 256);
    L.lse_col = off; off += pclip_align_up((size_t)B * 4, 256);
    L.slab_x = off; off += pclip_align_up(nT * B * d * 4, 256);
    L.slab_y = off; off += pclip_align_up(nT * B * d * 4, 256);
    L.total = off;
    return L;
}
size_t clip_step_fused_ws_bytes(int B, int d) { return clip_step_layout(B, d).total; }

// t == NULL: e_pep / e_rec are x, y as the towers return them (no normalise, no temperature; dt is not touched)
int clip_step_fused_launch(const float* e_pep, const float* e_rec, const float* t, int B, int d, float* loss, float* de_pep, float* de_rec,
                           float* dt, void* ws, void* stream) {
    const ClipStepWs L = clip_step_layout(B, d);
    char* w = (char*)ws;
    ClipStepArgs a;
    a.e_pep = e_pep; a.e_rec = e_rec; a.t = t; a.loss = loss; a.de_pep = de_pep; a.de_rec = de_rec; a.dt = dt;
    a.B = B; a.d = d; a.nT = (B + 31) / 32; a.scale = 0.5f / (float)B; a.raw = t != nullptr;
    a.counter = (unsigned*)(w + L.counter);
    a.part_row = (float2*)(w + L.part_row); a.part_col = (float2*)(w + L.part_col);
    a.diag = (float*)(w + L.diag); a.lse_row = (float*)(w + L.lse_row); a.lse_col = (float*)(w + L.lse_col);
    a.slab_x = (float*)(w + L.slab_x); a.slab_y = (float*)(w + L.slab_y);
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_cs_mu);
    ClipStepDevice* D = clip_step_device_locked();
    if (!D || D->cap <= 0 || !D->sync) return pclip_fail(PCLIP_ELAUNCH, "clip_step: the one-launch form is not available on the current device");
    a.counter = D->sync;
    a.stamps = (unsigned*)(w + L.counter);
    a.extra = (unsigned)g_cs_selftest_extra;
    // Two grids that wait at barriers must not share the GPU: each could hold part of the compute units and wait for
    // the rest for ever.  Launches on one stream follow each other anyway; when the stream changes, the new stream
    // first waits for everything enqueued so far on the one used last.  A stream under capture is left alone (an
    // event wait would pull the other stream into the capture): replaying such a graph beside other one-launch
    // steps of the same GPU is the caller's to avoid.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap == hipStreamCaptureStatusNone) {
        if (D->launched && D->last != st) {
            // the stream used last may have been destroyed since: its work then completes on its own, nothing to wait for
            if (hipEventRecord(D->chain, D->last) == hipSuccess) {
                if (hipStreamWaitEvent(st, D->chain, 0) != hipSuccess) return pclip_fail(PCLIP_ELAUNCH, "clip_step: hipStreamWaitEvent failed");
            } else {
                (void)hipGetLastError();
            }
        }
        D->launched = true;
        D->last = st;
    }
    hipLaunchKernelGGL(clip_step_kernel, dim3(a.nT * a.nT), dim3(256), CS_LDS_BYTES, st, a);   // the LDS limit was raised by the capacity query
    return pclip_check_launch("clip_step_kernel");
}

// ---- diagnostics of the barrier words -----------------------------------------------------
extern "C" int pclip_clip_step_sync_status(int* gave_up) {
    PCLIP_REQUIRE(gave_up, "clip_step_sync_status: null pointer");
    unsigned* sync = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_cs_mu);
        const ClipStepDevice* D = clip_step_device_locked();
        if (!D || !D->sync) { *gave_up = 0; return PCLIP_OK; }   // the one-launch form never runs on this device
        sync = D->sync;
    }
    unsigned words[4] = {0, 0, 0, 0};
    if (hipMemcpy(words, sync, sizeof(words), hipMemcpyDeviceToHost) != hipSuccess)
        return pclip_fail(PCLIP_ELAUNCH, "clip_step_sync_status: cannot read the barrier words");
    *gave_up = words[1] != 0u;
    return PCLIP_OK;
}

extern "C" int pclip_clip_step_sync_reset(void* stream) {
    std::lock_guard<std::mutex> lock(g_cs_mu);
    const ClipStepDevice* D = clip_step_device_locked();
    if (!D || !D->sync) return PCLIP_OK;
    if (hipMemsetAsync(D->sync, 0, 256, (hipStream_t)stream) != hipSuccess) return pclip_fail(PCLIP_ELAUNCH, "clip_step_sync_reset: memset failed");
    return PCLIP_OK;
}

extern "C" int pclip_clip_step_selftest_giveup(int on) {
    std::lock_guard<std::mutex> lock(g_cs_mu);
    g_cs_selftest_extra = on ? 1 : 0;
    return PCLIP_OK;
}
