// Microbenchmark: which ingredient of the FILIP score kernel's inner loop costs what?  A workgroup of 16 waves
// (four per SIMD) runs ITER "products".  The base product is a chain of 8 v_mfma_f32_32x32x16_bf16 followed by
// the key work on ITS result (16 shift-or, 16 max, 8 max3 = 40 vector instructions); flags add, one at a time
// or together:
//   LDSB  the eight B fragments come from LDS (8 ds_read_b128 per product, conflict-free) instead of registers
//   ATOM  one 64-bit LDS atomic max per product and lane
//   BARR  a workgroup barrier every fourth product
//   SALU  48 extra scalar instructions per product
//   PRIO  s_setprio by progress inside an epoch of four products
// Prints ns per product and SIMD (kernel time / ITER / 4).
//   hipcc -O3 --offload-arch=gfx950 stream_model.cpp -o stream_model && ./stream_model
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool LDSB, bool ATOM, bool BARR, bool SALU, bool PRIO>
__global__ __launch_bounds__(1024) void k(const bf16x8* __restrict__ in, int* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long long* col = reinterpret_cast<long long*>(smem + 65536);
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8 a[8], b[8];
    for (int s = 0; s < 8; ++s) { a[s] = in[(tid & 255) * 16 + s]; b[s] = in[(tid & 255) * 16 + 8 + s]; }
    for (int i = tid; i < 65536 / 16; i += 1024) reinterpret_cast<bf16x8*>(smem)[i] = in[i & 4095];
    for (int i = tid; i < 512; i += 1024) col[i] = 0;
    __syncthreads();
    int km[16];
    for (int r = 0; r < 16; ++r) km[r] = -r;
    int sacc = 0;
    for (int it = 0; it < iters; ++it) {
        if (PRIO) {
            if ((it & 3) == 0) __builtin_amdgcn_s_setprio(3);
            else if ((it & 3) == 1) __builtin_amdgcn_s_setprio(2);
            else if ((it & 3) == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        if (LDSB) {
            const unsigned char* base = smem + (it & 7) * 8192 + lane * 16;
#pragma unroll
            for (int s = 0; s < 8; ++s) b[s] = *reinterpret_cast<const bf16x8*>(base + 1024 * s);
        } else {
            asm volatile("" : "+v"(a[0]));
        }
        const f32x16 four = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], four, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc, 0, 0, 0);
        if (SALU) {
#pragma unroll
            for (int v = 0; v < 48; ++v) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sacc) : "s"(it) : "scc");
        }
        int key[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) key[r] = (int)(((unsigned)__float_as_int(acc[r]) << 7) | (unsigned)((it & 7) << 4 | r));
#pragma unroll
        for (int r = 0; r < 16; ++r) km[r] = max(km[r], key[r]);
        int cb = 0x80000000;
#pragma unroll
        for (int r = 0; r < 16; r += 2) cb = max(max(cb, key[r]), key[r + 1]);
        if (ATOM) atomicMax(&col[((it & 7) * 2 + (tid >> 9)) * 32 + (lane & 31)], ((long long)cb << 32) | lane);
        else km[0] = max(km[0], cb);
        if (BARR && (it & 3) == 3) __builtin_amdgcn_s_barrier();
    }
    int x = sacc;
    for (int r = 0; r < 16; ++r) x ^= km[r];
    out[blockIdx.x * 1024 + tid] = x;
}

// Two A tiles per wave: every B fragment read from LDS feeds two matrix instructions (two independent chains),
// the key work is done for both results.  NW waves per workgroup (12 = three per SIMD at <= 168 registers).
template <int NW, bool ATOM, bool BARR, bool SALU>
__global__ __launch_bounds__(NW * 64) void k2(const bf16x8* __restrict__ in, int* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long long* col = reinterpret_cast<long long*>(smem + 65536);
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8 a0[8], a1[8];
    for (int s = 0; s < 8; ++s) { a0[s] = in[(tid & 255) * 16 + s]; a1[s] = in[(tid & 255) * 16 + 8 + s]; }
    for (int i = tid; i < 65536 / 16; i += NW * 64) reinterpret_cast<bf16x8*>(smem)[i] = in[i & 4095];
    for (int i = tid; i < 512; i += NW * 64) col[i] = 0;
    __syncthreads();
    int km0[16], km1[16];
    for (int r = 0; r < 16; ++r) { km0[r] = -r; km1[r] = r; }
    int sacc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* base = smem + (it & 7) * 8192 + lane * 16;
        const f32x16 four = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
        f32x16 acc0 = four, acc1 = four;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(base + 1024 * s);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[s], b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s], b, acc1, 0, 0, 0);
        }
        if (SALU) {
#pragma unroll
            for (int v = 0; v < 96; ++v) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sacc) : "s"(it) : "scc");
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            int key[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) key[r] = (int)(((unsigned)__float_as_int(half ? acc1[r] : acc0[r]) << 7) | (unsigned)((it & 7) << 4 | r));
#pragma unroll
            for (int r = 0; r < 16; ++r) { if (half) km1[r] = max(km1[r], key[r]); else km0[r] = max(km0[r], key[r]); }
            int cb = 0x80000000;
#pragma unroll
            for (int r = 0; r < 16; r += 2) cb = max(max(cb, key[r]), key[r + 1]);
            if (ATOM) atomicMax(&col[((it & 7) * 2 + half) * 32 + (lane & 31)], ((long long)cb << 32) | lane);
            else km0[0] = max(km0[0], cb);
        }
        if (BARR && (it & 3) == 3) __builtin_amdgcn_s_barrier();
    }
    int x = sacc;
    for (int r = 0; r < 16; ++r) x ^= km0[r] ^ km1[r];
    out[blockIdx.x * NW * 64 + tid] = x;
}

template <int NW, bool ATOM, bool BARR, bool SALU>
static void run2(const char* name, bf16x8* in, int* out) {
    const int iters = 4000;
    (void)hipFuncSetAttribute((const void*)k2<NW, ATOM, BARR, SALU>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k2<NW, ATOM, BARR, SALU>), dim3(256), dim3(NW * 64), 65536 + 4096, 0, in, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / iters / (NW / 4) / 2;   // per product and SIMD
    printf("%-44s %6.1f ns per product and SIMD  (%4.0f TFLOP/s)\n", name, ns, 1024.0 * 8 * 32768 / ns * 1e-3);
}

template <bool LDSB, bool ATOM, bool BARR, bool SALU, bool PRIO>
static void run(const char* name, bf16x8* in, int* out) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)k<LDSB, ATOM, BARR, SALU, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<LDSB, ATOM, BARR, SALU, PRIO>), dim3(256), dim3(1024), 65536 + 4096, 0, in, out, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / iters / 4;
    printf("%-44s %6.1f ns per product and SIMD  (%4.0f TFLOP/s)\n", name, ns, 1024.0 * 8 * 32768 / ns * 1e-3);
}

int main() {
    bf16x8* in; int* out;
    hipMalloc(&in, 4096 * sizeof(bf16x8)); hipMemset(in, 0, 4096 * sizeof(bf16x8));
    hipMalloc(&out, 256 * 1024 * 4);
    run<false, false, false, false, false>("base (8 MFMA + 40 key ops)", in, out);
    run<true, false, false, false, false>("+ B fragments from LDS", in, out);
    run<false, true, false, false, false>("+ LDS atomic", in, out);
    run<false, false, true, false, false>("+ barrier every 4 products", in, out);
    run<false, false, false, true, false>("+ 48 scalar instructions", in, out);
    run<true, true, false, false, false>("+ LDS fragments + atomic", in, out);
    run<true, true, true, false, false>("+ LDS fragments + atomic + barrier", in, out);
    run<true, true, true, true, false>("+ all but priorities", in, out);
    run<true, true, true, true, true>("+ all", in, out);
    run<true, true, false, true, true>("all but the barrier", in, out);
    run2<12, false, false, false>("2 A tiles x 12 waves: LDS fragments", in, out);
    run2<12, true, true, true>("2 A tiles x 12 waves: all", in, out);
    run2<8, true, true, true>("2 A tiles x 8 waves: all", in, out);
    run2<16, true, true, true>("2 A tiles x 16 waves: all", in, out);
    return 0;
}
