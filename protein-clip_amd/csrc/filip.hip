// FILIP token-wise max similarity on bf16 matrix cores (v_mfma_f32_32x32x16_bf16), gfx950.
//
// Layout in HBM: each sequence is a [Tp][128] bf16 image (Tp = tokens rounded up to 32,
// width zero-padded to 128; padding rows and masked tokens are zero) plus one u32 of mask
// bits per 32-token tile.  The [B,B,TA,TB] similarity tensor of the reference is never
// formed: a workgroup owns one A sequence (its tiles live in registers as MFMA A operands),
// streams B sequences through an XOR-swizzled LDS image, and reduces each 32x32 tile to a
// running row maximum (elementwise, reduced across lanes once per pair) and a column
// maximum (over registers) as soon as the tile leaves the matrix core.
#include "pclip_common.h"

#define FD 128            // padded embedding width
#define ROWB (FD * 2)     // bytes per token row

// ---- prep: f32 [B,T,d] -> bf16 [B,Tp,128] (+ mask words) --------------------------------
// One block per 32-token tile of one sequence: eight lanes per token (widths that are a multiple of four), or
// wave w converts tokens w, w+4, ...
__global__ __launch_bounds__(256) void filip_prep_kernel(const float* __restrict__ h, const uint8_t* __restrict__ mask, int B, int T, int d,
                                                         int Tp, int normalise, const float* __restrict__ tsign,
                                                         uint16_t* __restrict__ hn, uint32_t* __restrict__ maskw) {
    const bool negate = tsign && tsign[0] < 0.f;
    const int ntile = Tp >> 5;
    // The hardware deals block b to XCD b % 8 (and to shader engine (b / 8) % 4 inside it): with tile = b % ntile an XCD
    // would always get the same tile position, and the late tiles of ragged sequences are padding.  Rotated by the
    // sequence, every XCD sees every position.
    const int seq = blockIdx.x / ntile;
    const int tile = (blockIdx.x - seq * ntile + seq) % ntile;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w == 0) {
        const int tok = tile * 32 + (lane & 31);
        const bool on = lane < 32 && tok < T && mask[(size_t)seq * T + tok] != 0;
        const unsigned long long bal = __ballot(on);
        if (lane == 0) maskw[(size_t)seq * ntile + tile] = (uint32_t)bal;
    }
    if ((d & 3) == 0 && ((uintptr_t)h & 15) == 0) {
        // eight lanes per token, sixteen columns each: the whole tile in one pass, four 16-byte loads per thread in flight
        const int r = threadIdx.x >> 3, l8 = threadIdx.x & 7;
        const int tok = tile * 32 + r;
        // the row is requested together with its mask byte (no load waits for another); a masked row is dropped afterwards
        const uint8_t mbyte = tok < T ? mask[(size_t)seq * T + tok] : (uint8_t)0;
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 16 * l8 + 4 * u;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tok < T && k < d) v[u] = *reinterpret_cast<const float4*>(h + ((size_t)seq * T + tok) * d + k);
        }
        const bool live = mbyte != 0;
        if (!live) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (normalise) {
            float ss = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) ss += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
            ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
            const float nrm = sqrtf(ss);
            if (live) {   // no epsilon, as in the reference (an all-zero valid token divides by zero there too)
#pragma unroll
                for (int u = 0; u < 4; ++u) { v[u].x /= nrm; v[u].y /= nrm; v[u].z /= nrm; v[u].w /= nrm; }
            }
        }
        const float sg = negate ? -1.f : 1.f;
        uint32_t pk[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const __bf16 b0 = (__bf16)(sg * v[u].x), b1 = (__bf16)(sg * v[u].y), b2 = (__bf16)(sg * v[u].z), b3 = (__bf16)(sg * v[u].w);
            pk[2 * u] = (uint32_t)__builtin_bit_cast(uint16_t, b0) | ((uint32_t)__builtin_bit_cast(uint16_t, b1) << 16);
            pk[2 * u + 1] = (uint32_t)__builtin_bit_cast(uint16_t, b2) | ((uint32_t)__builtin_bit_cast(uint16_t, b3) << 16);
        }
        uint4* dst = reinterpret_cast<uint4*>(hn + ((size_t)seq * Tp + tok) * FD + 16 * l8);
        dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        return;
    }
    for (int r = w; r < 32; r += 4) {
        const int tok = tile * 32 + r;
        const bool live = tok < T && mask[(size_t)seq * T + tok] != 0;
        float v0 = 0.f, v1 = 0.f;
        if (live) {
            const float* hr = h + ((size_t)seq * T + tok) * d;
            const int k = 2 * lane;
            if (k < d) v0 = hr[k];
            if (k + 1 < d) v1 = hr[k + 1];
            if (normalise) {
                const float ss = wave_sum(v0 * v0 + v1 * v1);
                const float nrm = sqrtf(ss);
                v0 = v0 / nrm;
                v1 = v1 / nrm;
            }
            if (negate) {
                v0 = -v0;
                v1 = -v1;
            }
        }
        const __bf16 b0 = (__bf16)v0, b1 = (__bf16)v1;
        const uint32_t packed = (uint32_t)__builtin_bit_cast(uint16_t, b0) | ((uint32_t)__builtin_bit_cast(uint16_t, b1) << 16);
        reinterpret_cast<uint32_t*>(hn + ((size_t)seq * Tp + tok) * FD)[lane] = packed;
    }
}

// Data that is read once (or written and not read again soon) goes past the caches' keep lists: the partial gradients
// of the gathers are 268 MB per side at the benchmark shape, more than the memory-side cache holds, and reading them the
// ordinary way evicts what the next gather needs (partner images, arg tables, lists): gathers 336 -> 326 us.
typedef float f32x2n __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 stream_load2(const float* p) {
    const f32x2n v = __builtin_nontemporal_load(reinterpret_cast<const f32x2n*>(p));
    return make_float2(v.x, v.y);
}

// ---- prep backward: dn [B,Tp,128] f32 -> dh [B,T,d] f32 --------------------------------
// Masked and padding tokens carry a zero dn (the score backward never touches them), so
// their dh comes out zero without consulting the mask.
__global__ __launch_bounds__(256) void filip_prep_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dn, int nslab,
                                                             size_t slab_stride, size_t row_stride, const uint32_t* __restrict__ mw, int B, int T, int d,
                                                             int Tp, int normalise, const float* __restrict__ tsign, int accumulate,
                                                             float* __restrict__ dh) {
    // dn is the sum of `nslab` slabs `slab_stride` floats apart, rows `row_stride` floats apart (the partial gradients of the
    // windowed backward); with mask words given, rows of padding tokens are taken as zero
    // without being read.
    const bool negate = tsign && tsign[0] < 0.f;
    const int lane = threadIdx.x & 63;
    // groups of eight blocks (32 rows) go to one shader engine: the group's place inside its sequence is rotated by the
    // sequence, so that an engine does not always get the same token range (the late ones are padding at ragged lengths)
    size_t blk = blockIdx.x;
    if ((T & 31) == 0) {
        const size_t gps = (size_t)(T >> 5), grp = blk >> 3, s = grp / gps;
        blk = ((s * gps + (grp - s * gps + s) % gps) << 3) | (blk & 7);
    }
    const size_t row = blk * 4 + (threadIdx.x >> 6);
    if (row >= (size_t)B * T) return;
    const int seq = (int)(row / T), tok = (int)(row - (size_t)seq * T);
    const float* hr = h + row * d;
    const float* gr = dn + ((size_t)seq * Tp + tok) * row_stride;
    float* out = dh + row * d;
    const int k = 2 * lane;
    const bool live = !mw || ((mw[(size_t)seq * (Tp >> 5) + (tok >> 5)] >> (tok & 31)) & 1u);
    float g0 = 0.f, g1 = 0.f, v0 = 0.f, v1 = 0.f;
    if (k < d) v0 = hr[k];
    if (k + 1 < d) v1 = hr[k + 1];
    if (live) {   // dn rows are 128 floats wide whatever d is: one 8-byte load per slab, eight in flight
        int s = 0;
        for (; s + 8 <= nslab; s += 8) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = stream_load2(gr + (size_t)(s + u) * slab_stride + k);
#pragma unroll
            for (int u = 0; u < 8; ++u) { g0 += v[u].x; g1 += v[u].y; }
        }
        for (; s < nslab; ++s) {
            const float2 v = stream_load2(gr + (size_t)s * slab_stride + k);
            g0 += v.x; g1 += v.y;
        }
        if (k >= d) g0 = 0.f;
        if (k + 1 >= d) g1 = 0.f;
    }
    const float sgn = negate ? -1.f : 1.f;
    float o0 = g0 * sgn, o1 = g1 * sgn;
    if (normalise) {
        const float ss = wave_sum(v0 * v0 + v1 * v1);
        const float dot = wave_sum(v0 * g0 + v1 * g1);
        if (ss > 0.f) {
            const float inv = 1.f / sqrtf(ss);
            const float ndg = dot * inv * inv;  // (h.g)/|h|^2
            o0 = (g0 - v0 * ndg) * inv * sgn;
            o1 = (g1 - v1 * ndg) * inv * sgn;
        }
    }
    if (k + 1 < d && (d & 1) == 0 && !accumulate && ((uintptr_t)dh & 7) == 0) {   // the usual case: one 8-byte streaming store per lane (43 instead of 48 us)
        f32x2n ov;
        ov.x = o0; ov.y = o1;
        __builtin_nontemporal_store(ov, reinterpret_cast<f32x2n*>(out + k));
        return;
    }
    if (k < d) out[k] = accumulate ? out[k] + o0 : o0;
    if (k + 1 < d) out[k + 1] = accumulate ? out[k + 1] + o1 : o1;
}

// The same for the usual case (d a multiple of 4, 16-byte aligned rows, dh overwritten): half a wave per row with
// 16-byte loads and stores, all slabs of up to four in flight at once, so a wave has twice the bytes on their way.
typedef float f32x4n __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void filip_prep_bwd_rows2_kernel(const float* __restrict__ h, const float* __restrict__ dn, int nslab,
                                                                   size_t slab_stride, size_t row_stride, const uint32_t* __restrict__ mw, int B, int T,
                                                                   int d, int Tp, int normalise, const float* __restrict__ tsign,
                                                                   float* __restrict__ dh) {
    const bool negate = tsign && tsign[0] < 0.f;
    const int lane = threadIdx.x & 63, l32 = lane & 31;
    size_t blk = blockIdx.x;   // eight rows per block, four blocks per 32-token group; groups rotated as above
    if ((T & 31) == 0) {
        const size_t gps = (size_t)(T >> 5), grp = blk >> 2, s = grp / gps;
        blk = ((s * gps + (grp - s * gps + s) % gps) << 2) | (blk & 3);
    }
    const size_t row = blk * 8 + (threadIdx.x >> 5);
    if (row >= (size_t)B * T) return;
    const int seq = (int)(row / T), tok = (int)(row - (size_t)seq * T);
    const int k = 4 * l32;
    const bool in = k < d;
    const bool live = !mw || ((mw[(size_t)seq * (Tp >> 5) + (tok >> 5)] >> (tok & 31)) & 1u);
    f32x4n v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
    if (in) v = *reinterpret_cast<const f32x4n*>(h + row * d + k);
    if (live && in) {
        const f32x4n* gr = reinterpret_cast<const f32x4n*>(dn + ((size_t)seq * Tp + tok) * row_stride + k);
        const size_t ss4 = slab_stride >> 2;
        int s = 0;
        for (; s + 4 <= nslab; s += 4) {
            f32x4n p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = __builtin_nontemporal_load(gr + (size_t)(s + u) * ss4);
#pragma unroll
            for (int u = 0; u < 4; ++u) g += p[u];
        }
        for (; s < nslab; ++s) g += __builtin_nontemporal_load(gr + (size_t)s * ss4);
    }
    const float sgn = negate ? -1.f : 1.f;
    f32x4n o = g * sgn;
    if (normalise) {
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        float dot = v.x * g.x + v.y * g.y + v.z * g.z + v.w * g.w;
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) {   // the 32 lanes of the row
            ss += __shfl_xor(ss, m);
            dot += __shfl_xor(dot, m);
        }
        if (ss > 0.f) {
            const float inv = 1.f / sqrtf(ss);
            const float ndg = dot * inv * inv;  // (h.g)/|h|^2
            o = (g - v * ndg) * (inv * sgn);
        }
    }
    if (in) __builtin_nontemporal_store(o, reinterpret_cast<f32x4n*>(dh + row * d + k));
}

// internal: prep backward over a sum of slabs (used by the fused backward of filip_bwd.hip)
int filip_prep_bwd_launch(const float* h, const float* dn, int nslab, size_t slab_stride, size_t row_stride, const uint32_t* mw, int B, int T, int d,
                          int normalise, const float* tsign, int accumulate, float* dh, void* stream) {
    const int Tp = (T + 31) / 32 * 32;
    const size_t rows = (size_t)B * T;
    if ((d & 3) == 0 && !accumulate && (slab_stride & 3) == 0 && (row_stride & 3) == 0 && (((uintptr_t)h | (uintptr_t)dn | (uintptr_t)dh) & 15) == 0) {
        hipLaunchKernelGGL(filip_prep_bwd_rows2_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, h, dn, nslab, slab_stride,
                           row_stride, mw, B, T, d, Tp, normalise, tsign, dh);
        return pclip_check_launch("filip_prep_bwd_rows2_kernel");
    }
    hipLaunchKernelGGL(filip_prep_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, dn, nslab, slab_stride, row_stride, mw,
                       B, T, d, Tp, normalise, tsign, accumulate, dh);
    return pclip_check_launch("filip_prep_bwd_kernel");
}

// ---- scores ------------------------------------------------------------------------------
struct FilipArgs {
    const uint16_t* hnA; const uint32_t* mwA; int Br, TAp;
    const uint16_t* hnB; const uint32_t* mwB; int Bc, TBp;
    const float* t;
    float* simA; float* simB;
    uint16_t* argA; uint16_t* argB;
    int jchunk;  // B sequences per workgroup
    int ldc, col0;   // the outputs are [Br, ldc(, T)] arrays of which this launch fills the columns col0 .. col0 + Bc - 1
};

// LDS: B image [TBp][256 B] swizzled | colpart float2 [4][256] | rsum float[4] | red float[4]
#define LDS_IMG 0
#define LDS_COLPART (256 * ROWB)
#define LDS_RSUM (LDS_COLPART + 4 * 256 * 8)
#define LDS_RED (LDS_RSUM + 16)
#define LDS_TOTAL (LDS_RED + 32)

__global__ __launch_bounds__(256, 2) void filip_sim_kernel(FilipArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* colpart = reinterpret_cast<float2*>(smem + LDS_COLPART);
    float* rsum = reinterpret_cast<float*>(smem + LDS_RSUM);
    float* red = reinterpret_cast<float*>(smem + LDS_RED);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int i = blockIdx.x % p.Br;
    const int chunk = blockIdx.x / p.Br;
    const int j_begin = chunk * p.jchunk, j_end = min(p.Bc, j_begin + p.jchunk);
    const int ntA = p.TAp >> 5, ntB = p.TBp >> 5;
    const float inv_t = 1.f / fabsf(p.t[0]);

    // this wave's A tiles: slots 0,1 <-> tiles w, w+4
    bf16x8 afr[2][8];
    uint32_t wA[2];
    int nA = 0;
    for (int tt = 0; tt < ntA; ++tt) nA += __popc(p.mwA[(size_t)i * ntA + tt]);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int at = w + 4 * sl;
        wA[sl] = (at < ntA) ? p.mwA[(size_t)i * ntA + at] : 0u;
        const uint16_t* src = p.hnA + ((size_t)i * p.TAp + (size_t)min(at, ntA - 1) * 32 + c) * FD + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) afr[sl][s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
    }

    for (int j = j_begin; j < j_end; ++j) {
        // ---- stage hB[j] into the swizzled LDS image ----
        int nB = 0, last_tile = -1;
        for (int tt = 0; tt < ntB; ++tt) {
            const uint32_t wd = p.mwB[(size_t)j * ntB + tt];
            nB += __popc(wd);
            if (wd) last_tile = tt;
        }
        __syncthreads();  // previous pair done with the image and colpart
        {
            const int nchunks = (last_tile + 1) * 32 * 16;
            const uint4* src = reinterpret_cast<const uint4*>(p.hnB + (size_t)j * p.TBp * FD);
            for (int g = tid; g < nchunks; g += 256) {
                const int row = g >> 4, ch = g & 15;
                *reinterpret_cast<uint4*>(smem + LDS_IMG + row * ROWB + ((ch ^ (row & 15)) << 4)) = src[g];
            }
        }
        __syncthreads();

        float M[2][16];
        int MI[2][16];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int r = 0; r < 16; ++r) { M[sl][r] = -FLT_MAX; MI[sl][r] = 0; }

        for (int bt = 0; bt <= last_tile; ++bt) {
            const uint32_t wB = p.mwB[(size_t)j * ntB + bt];
            if (wB == 0) continue;
            bf16x8 bfr[8];
            {
                const int row = bt * 32 + c;
                const unsigned char* base = smem + LDS_IMG + row * ROWB;
#pragma unroll
                for (int s = 0; s < 8; ++s) bfr[s] = *reinterpret_cast<const bf16x8*>(base + (((2 * s + h) ^ (c & 15)) << 4));
            }
            float cbest = -FLT_MAX;
            int carg = 0;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (wA[sl] == 0) continue;
                const int at = w + 4 * sl;
                f32x16 acc;
                if (wA[sl] == 0xffffffffu && wB == 0xffffffffu) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                } else {
                    const float colbias = ((wB >> c) & 1u) ? 0.f : -FLT_MAX;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = ((wA[sl] >> acc_row(r, h)) & 1u) ? colbias : -FLT_MAX;
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[sl][s], bfr[s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool up = acc[r] > M[sl][r];
                    M[sl][r] = up ? acc[r] : M[sl][r];
                    MI[sl][r] = up ? bt : MI[sl][r];
                    const bool upc = acc[r] > cbest;
                    cbest = upc ? acc[r] : cbest;
                    carg = upc ? (at * 32 + acc_row(r, h)) : carg;
                }
            }
            {
                const float ob = __shfl_xor(cbest, 32);
                const int oa = __shfl_xor(carg, 32);
                if (ob > cbest || (ob == cbest && oa < carg)) { cbest = ob; carg = oa; }
                if (h == 0) colpart[w * 256 + bt * 32 + c] = make_float2(cbest, __int_as_float(carg));
            }
        }

        // ---- rows: reduce each register across the 32 lanes of its half (value + arg) ----
        float rs = 0.f;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int at = w + 4 * sl;
            if (wA[sl] == 0) {  // tile of padding only: its arg-max slots are defined (zero), nothing else to do
                if (p.argA && at < ntA && c < 16) p.argA[((size_t)i * p.ldc + p.col0 + j) * p.TAp + at * 32 + acc_row(c, h)] = 0;
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = M[sl][r];
                int b = MI[sl][r] * 32 + c;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    const float ov = __shfl_xor(v, o);
                    const int ob = __shfl_xor(b, o);
                    if (ov > v || (ov == v && ob < b)) { v = ov; b = ob; }
                }
                const int arow = acc_row(r, h);
                const bool valid = (wA[sl] >> arow) & 1u;
                rs += valid ? v : 0.f;
                if (p.argA && c == r) p.argA[((size_t)i * p.ldc + p.col0 + j) * p.TAp + at * 32 + arow] = valid ? (uint16_t)b : (uint16_t)0;
            }
        }
        rs += __shfl_xor(rs, 32);
        if (lane == 0) rsum[w] = rs;
        __syncthreads();

        // ---- columns: merge the four waves' partial maxima, masked sum ----
        float cv = 0.f;
        if (tid < p.TBp) {
            const int bt = tid >> 5;
            const uint32_t wB = p.mwB[(size_t)j * ntB + bt];
            const bool valid = (wB >> (tid & 31)) & 1u;
            float best = -FLT_MAX;
            int arg = 0;
            if (valid) {
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float2 q = colpart[ww * 256 + tid];
                    const int qa = __float_as_int(q.y);
                    if (q.x > best || (q.x == best && qa < arg)) { best = q.x; arg = qa; }
                }
                cv = best;
            }
            if (p.argB) p.argB[((size_t)i * p.ldc + p.col0 + j) * p.TBp + tid] = (uint16_t)arg;
        }
        cv = wave_sum(cv);
        if (lane == 0) red[w] = cv;
        __syncthreads();
        if (tid == 0) {
            const float sa = rsum[0] + rsum[1] + rsum[2] + rsum[3];
            const float sb = red[0] + red[1] + red[2] + red[3];
            p.simA[(size_t)i * p.ldc + p.col0 + j] = sa / fmaxf((float)nA, 1e-6f) * inv_t;
            p.simB[(size_t)i * p.ldc + p.col0 + j] = sb / fmaxf((float)nB, 1e-6f) * inv_t;
        }
    }
}

// ---- host side -------------------------------------------------------------------------
static int filip_dims_ok(const char* who, int B, int T, int d) {
    PCLIP_REQUIRE(B >= 1 && T >= 1 && T <= PCLIP_FILIP_MAX_T_LONG, "%s: B=%d T=%d (need 1 <= T <= %d)", who, B, T, PCLIP_FILIP_MAX_T_LONG);
    PCLIP_REQUIRE(d >= 1 && d <= PCLIP_FILIP_MAX_D, "%s: d=%d (need 1 <= d <= %d)", who, d, PCLIP_FILIP_MAX_D);
    return PCLIP_OK;
}

extern "C" int pclip_filip_prep(const float* h, const uint8_t* mask, int B, int T, int d, int normalise, const float* tsign, uint16_t* hn,
                                uint32_t* maskw, void* stream) {
    PCLIP_REQUIRE(h && mask && hn && maskw, "filip_prep: null pointer");
    if (int rc = filip_dims_ok("filip_prep", B, T, d)) return rc;
    const int Tp = (T + 31) / 32 * 32;
    hipLaunchKernelGGL(filip_prep_kernel, dim3(B * (Tp / 32)), dim3(256), 0, (hipStream_t)stream, h, mask, B, T, d, Tp, normalise, tsign, hn,
                       maskw);
    return pclip_check_launch("filip_prep_kernel");
}

extern "C" int pclip_filip_prep_bwd(const float* h, const float* dn, int B, int T, int d, int normalise, const float* tsign, int accumulate,
                                    float* dh, void* stream) {
    PCLIP_REQUIRE(h && dn && dh, "filip_prep_bwd: null pointer");
    if (int rc = filip_dims_ok("filip_prep_bwd", B, T, d)) return rc;
    return filip_prep_bwd_launch(h, dn, 1, 0, FD, nullptr, B, T, d, normalise, tsign, accumulate, dh, stream);
}

extern "C" size_t pclip_filip_sim_fwd_workspace_bytes(int Br, int TA, int Bc, int TB, int unit_norm) {
    if (Br < 1 || Bc < 1 || TA < 1 || TB < 1 || TA > PCLIP_FILIP_MAX_T_LONG || TB > PCLIP_FILIP_MAX_T_LONG) return 0;
    if (!unit_norm || (TA <= PCLIP_FILIP_MAX_T && TB <= PCLIP_FILIP_MAX_T)) return 0;
    return filip_sim_long_stream_ws_bytes(Br, (TA + 31) / 32 * 32, Bc, (TB + 31) / 32 * 32);
}

extern "C" int pclip_filip_sim_fwd_cols(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA, const uint16_t* hnB, const uint32_t* maskwB,
                                        int Bc, int TB, const float* t, int unit_norm, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB,
                                        int col_offset, int Bc_total, void* stream) {
    return pclip_filip_sim_fwd_ws(hnA, maskwA, Br, TA, hnB, maskwB, Bc, TB, t, unit_norm, sim_A, sim_B, argA, argB, col_offset, Bc_total, nullptr, 0,
                                  stream);
}

extern "C" int pclip_filip_sim_fwd_ws(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA, const uint16_t* hnB, const uint32_t* maskwB,
                                      int Bc, int TB, const float* t, int unit_norm, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB,
                                      int col_offset, int Bc_total, void* ws, size_t ws_bytes, void* stream) {
    PCLIP_REQUIRE(hnA && maskwA && hnB && maskwB && t && sim_A && sim_B, "filip_sim_fwd: null pointer");
    PCLIP_REQUIRE((argA == nullptr) == (argB == nullptr), "filip_sim_fwd: pass both argA and argB or neither");
    if (int rc = filip_dims_ok("filip_sim_fwd", Br, TA, FD)) return rc;
    if (int rc = filip_dims_ok("filip_sim_fwd", Bc, TB, FD)) return rc;
    PCLIP_REQUIRE(col_offset >= 0 && Bc_total >= 1 && col_offset + Bc <= Bc_total, "filip_sim_fwd: columns %d .. %d of %d", col_offset,
                  col_offset + Bc - 1, Bc_total);
    const int TAp = (TA + 31) / 32 * 32, TBp = (TB + 31) / 32 * 32;
    if (TA > PCLIP_FILIP_MAX_T || TB > PCLIP_FILIP_MAX_T) {
        // unit-norm operands and a workspace: 256-token segments on the streaming kernel; else the general kernel
        if (unit_norm && ws) {
            const size_t need = pclip_filip_sim_fwd_workspace_bytes(Br, TA, Bc, TB, unit_norm);
            PCLIP_REQUIRE(ws_bytes >= need, "filip_sim_fwd: workspace %zu < %zu bytes", ws_bytes, need);
            PCLIP_REQUIRE((uintptr_t)ws % 16 == 0, "filip_sim_fwd: the workspace must be 16-byte aligned");
            return filip_sim_long_stream_launch(hnA, maskwA, Br, TAp, hnB, maskwB, Bc, TBp, t, sim_A, sim_B, argA, argB, col_offset, Bc_total, ws,
                                                stream);
        }
        return filip_sim_long_launch(hnA, maskwA, Br, TAp, hnB, maskwB, Bc, TBp, t, sim_A, sim_B, argA, argB, col_offset, Bc_total, stream);
    }
    if (unit_norm)
        return filip_sim_stream_launch(hnA, maskwA, Br, TAp, hnB, maskwB, Bc, TBp, t, sim_A, sim_B, argA, argB, col_offset, Bc_total,
                                       unit_norm >= 2 ? unit_norm : 0, stream);
    FilipArgs a;
    a.hnA = hnA; a.mwA = maskwA; a.Br = Br; a.TAp = TAp;
    a.hnB = hnB; a.mwB = maskwB; a.Bc = Bc; a.TBp = TBp;
    a.t = t; a.simA = sim_A; a.simB = sim_B; a.argA = argA; a.argB = argB;
    a.ldc = Bc_total; a.col0 = col_offset;
    // enough workgroups to fill 256 CUs twice over several times, but at least 8 B sequences each
    int nchunk = (2048 + Br - 1) / Br;
    if (nchunk < 1) nchunk = 1;
    int jchunk = (Bc + nchunk - 1) / nchunk;
    if (jchunk < 8) jchunk = Bc < 8 ? Bc : 8;
    nchunk = (Bc + jchunk - 1) / jchunk;
    a.jchunk = jchunk;
    static PclipLdsLimit lds_limit;
    if (int rc = pclip_ensure_lds(lds_limit, (const void*)filip_sim_kernel, LDS_TOTAL, "filip_sim_kernel")) return rc;
    hipLaunchKernelGGL(filip_sim_kernel, dim3(Br * nchunk), dim3(256), LDS_TOTAL, (hipStream_t)stream, a);
    return pclip_check_launch("filip_sim_kernel");
}

extern "C" int pclip_filip_sim_fwd(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA, const uint16_t* hnB, const uint32_t* maskwB,
                                   int Bc, int TB, const float* t, int unit_norm, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB,
                                   void* stream) {
    return pclip_filip_sim_fwd_cols(hnA, maskwA, Br, TA, hnB, maskwB, Bc, TB, t, unit_norm, sim_A, sim_B, argA, argB, 0, Bc, stream);
}
