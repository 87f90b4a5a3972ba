// Microbenchmark: the FILIP score kernel's inner loop with the matrix chain of one product issued BETWEEN the key
// instructions of the product before it, inside one wave.  A wave holds two A tiles (two accumulators); per B tile it
// runs the chain of A0 interleaved with the keys of A1's previous result, then the chain of A1 interleaved with the
// keys of A0's result; every B fragment register is reloaded from LDS right behind its last use (eight matrix
// instructions before its next one).  Compared with the plain order (both chains, then both key sets) of
// stream_model.cpp's k2.  NW waves per workgroup (8 = two per SIMD, 12 = three).
//   hipcc -O3 --offload-arch=gfx950 pipelined_model.cpp -o pipelined_model && ./pipelined_model
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// the key work on two registers of a finished accumulator: 2 shift-or, 2 max, 1 max3 (5 vector instructions)
#define KEYS2(acc, km, cb, r, tag)                                                        \
    {                                                                                       \
        const int k0 = (int)(((unsigned)__float_as_int(acc[r]) << 7) | (unsigned)(tag | (r)));       \
        const int k1 = (int)(((unsigned)__float_as_int(acc[(r) + 1]) << 7) | (unsigned)(tag | ((r) + 1))); \
        asm volatile("v_max_i32 %0, %0, %1" : "+v"(km[r]) : "v"(k0));                        \
        asm volatile("v_max_i32 %0, %0, %1" : "+v"(km[(r) + 1]) : "v"(k1));                  \
        cb = max(max(cb, k0), k1);                                                          \
    }

template <int NW, bool PIPE, bool ATOM, bool BARR>
__global__ __launch_bounds__(NW * 64) void k3(const bf16x8* __restrict__ in, int* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long long* col = reinterpret_cast<long long*>(smem + 65536);
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8 a0[8], a1[8], b[8];
    for (int s = 0; s < 8; ++s) { a0[s] = in[(tid & 255) * 16 + s]; a1[s] = in[(tid & 255) * 16 + 8 + s]; }
    for (int i = tid; i < 65536 / 16; i += NW * 64) reinterpret_cast<bf16x8*>(smem)[i] = in[i & 4095];
    for (int i = tid; i < 512; i += NW * 64) col[i] = 0;
    __syncthreads();
    int km0[16], km1[16];
    for (int r = 0; r < 16; ++r) { km0[r] = -r; km1[r] = r; }
    const f32x16 four = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
    f32x16 acc0 = four, acc1 = four;
    {
        const unsigned char* base = smem + lane * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = *reinterpret_cast<const bf16x8*>(base + 1024 * s);
    }
    for (int it = 0; it < iters; ++it) {
        typedef __attribute__((address_space(3))) const bf16x8 lds_frag;
        lds_frag* next = (lds_frag*)(smem + ((it + 1) & 7) * 8192 + lane * 16);
        const int tag = (it & 7) << 4;
        if (PIPE) {
            // chain of A0 for this tile || keys of A1's result of the tile before
            int cb1 = 0x80000000;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc0 = MFMA(a0[s], b[s], s == 0 ? four : acc0);
                __builtin_amdgcn_sched_barrier(0);
                KEYS2(acc1, km1, cb1, 2 * s, tag);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ATOM) atomicMax(&col[((it & 7) * 2 + 1) * 32 + (lane & 31)], ((long long)cb1 << 32) | lane);
            else km1[0] = max(km1[0], cb1);
            // chain of A1 || keys of A0's result; each fragment register reloaded behind its last use
            int cb0 = 0x80000000;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc1 = MFMA(a1[s], b[s], s == 0 ? four : acc1);
                __builtin_amdgcn_sched_barrier(0);
                b[s] = next[64 * s];
                KEYS2(acc0, km0, cb0, 2 * s, tag);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ATOM) atomicMax(&col[((it & 7) * 2) * 32 + (lane & 31)], ((long long)cb0 << 32) | lane);
            else km0[0] = max(km0[0], cb0);
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc0 = MFMA(a0[s], b[s], s == 0 ? four : acc0);
                acc1 = MFMA(a1[s], b[s], s == 0 ? four : acc1);
                b[s] = next[64 * s];
            }
            int cb0 = 0x80000000, cb1 = 0x80000000;
#pragma unroll
            for (int s = 0; s < 8; ++s) KEYS2(acc0, km0, cb0, 2 * s, tag);
#pragma unroll
            for (int s = 0; s < 8; ++s) KEYS2(acc1, km1, cb1, 2 * s, tag);
            if (ATOM) {
                atomicMax(&col[((it & 7) * 2) * 32 + (lane & 31)], ((long long)cb0 << 32) | lane);
                atomicMax(&col[((it & 7) * 2 + 1) * 32 + (lane & 31)], ((long long)cb1 << 32) | lane);
            } else {
                km0[0] = max(km0[0], cb0);
                km1[0] = max(km1[0], cb1);
            }
        }
        if (BARR && (it & 3) == 3) __builtin_amdgcn_s_barrier();
    }
    int x = 0;
    for (int r = 0; r < 16; ++r) x ^= km0[r] ^ km1[r] ^ __float_as_int(acc0[r]) ^ __float_as_int(acc1[r]);
    out[blockIdx.x * NW * 64 + tid] = x;
}

template <int NW, bool PIPE, bool ATOM, bool BARR>
static void run(const char* name, bf16x8* in, int* out) {
    const int iters = 4000;
    (void)hipFuncSetAttribute((const void*)k3<NW, PIPE, ATOM, BARR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k3<NW, PIPE, ATOM, BARR>), dim3(256), dim3(NW * 64), 65536 + 4096, 0, in, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / iters / (NW / 4) / 2;   // per product and SIMD
    printf("%-56s %6.1f ns per product and SIMD  (%4.0f TFLOP/s)\n", name, ns, 1024.0 * 8 * 32768 / ns * 1e-3);
}

int main() {
    bf16x8* in; int* out;
    (void)hipMalloc(&in, 4096 * sizeof(bf16x8)); (void)hipMemset(in, 0, 4096 * sizeof(bf16x8));
    (void)hipMalloc(&out, 256 * 1024 * 4);
    run<8, false, false, false>("8 waves x 2 tiles, plain order", in, out);
    run<8, true, false, false>("8 waves x 2 tiles, chain || previous keys", in, out);
    run<8, false, true, true>("8 waves x 2 tiles, plain, atomic + barrier", in, out);
    run<8, true, true, true>("8 waves x 2 tiles, pipelined, atomic + barrier", in, out);
    run<12, false, true, true>("12 waves x 2 tiles, plain, atomic + barrier", in, out);
    run<12, true, true, true>("12 waves x 2 tiles, pipelined, atomic + barrier", in, out);
    run<4, true, true, true>("4 waves x 2 tiles (one per SIMD), pipelined, atomic", in, out);
    return 0;
}
