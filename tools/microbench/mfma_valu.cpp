// Microbenchmark: how much plain vector work rides beside a chain of v_mfma_f32_32x32x16_bf16?
// (Mode 3: the same product as sixteen v_mfma_f32_16x16x32_bf16 in four independent chains.)
// Each wave runs ITER iterations of { 8 dependent MFMAs ; NV integer vector ops on 16 registers
// of the PREVIOUS iteration's result }, either one after the other (mode 0) or interleaved by
// sched_group_barrier, VPG vector ops after every MFMA (mode 1).  Prints shader cycles per
// iteration for 1, 2 and 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 mfma_valu.cpp -o mfma_valu && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE, int VPG>
__global__ __launch_bounds__(1024) void k(const bf16x8* __restrict__ in, int* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
    bf16x8 a[8], b[8];
    for (int s = 0; s < 8; ++s) { a[s] = in[threadIdx.x * 16 + s]; b[s] = in[threadIdx.x * 16 + 8 + s]; }
    f32x16 prev = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int km[16];
    for (int r = 0; r < 16; ++r) km[r] = -r;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const f32x16 two = {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2};
        asm volatile("" : "+v"(a[0]));   // the operands count as changed: the chain stays in the loop
        f32x16 acc;
        if (MODE == 3) {
            // the same 32 x 32 x 128 product as four 16 x 16 tiles of v_mfma_f32_16x16x32_bf16: four independent chains of
            // four instructions (two A and two B fragments per k-step of 32), 16 result registers as before
            f32x4 c[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = f32x4{2, 2, 2, 2};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * s + (q >> 1)], b[2 * s + (q & 1)], c[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * q + e] = c[q][e];
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], two, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc, 0, 0, 0);
        }
        // NV vector ops on the previous result: rounds of (and_or, max) over the 16 registers
#pragma unroll
        for (int v = 0; v < NV / 2; ++v) {
            const int r = v & 15;
            km[r] = max(km[r], (__float_as_int(MODE >= 2 ? acc[r] : prev[r]) & ~7) | (it & 7) | (v >> 4));
        }
        if (MODE == 1) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, MODE == 3 ? 16 : 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        }
        prev = acc;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int x = 0;
    for (int r = 0; r < 16; ++r) x ^= km[r] ^ __float_as_int(prev[r]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NV, int MODE, int VPG>
static void run(const char* name, bf16x8* in, int* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {
        const int threads = wps * 4 * 64;   // waves per SIMD x 4 SIMDs
        hipLaunchKernelGGL((k<NV, MODE, VPG>), dim3(256), dim3(threads), 0, 0, in, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NV, MODE, VPG>), dim3(256), dim3(threads), 0, 0, in, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const int nw = 256 * threads / 64;
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += v;
        // cycles per iteration per WAVE, and per SIMD per (wave-iteration) = the throughput figure
        const double ns_per_product = ms * 1e6 / iters / wps;   // per SIMD
        printf("%-28s waves/SIMD %d: %7.1f counts/iter/wave  %6.1f counts and %6.1f ns per product per SIMD  (%5.0f TFLOP/s, counter %.2f GHz)\n", name, wps,
               s / nw / iters, s / nw / iters / wps, ns_per_product, 1024.0 * 8 * 32768 / ns_per_product * 1e-3, s / nw / (ms * 1e6));
    }
}

int main() {
    bf16x8* in; int* out; unsigned long long* cyc;
    hipMalloc(&in, 1024 * 16 * sizeof(bf16x8)); hipMemset(in, 0, 1024 * 16 * sizeof(bf16x8));
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    run<0, 0, 0>("8 MFMA only", in, out, cyc);
    run<32, 0, 0>("8 MFMA then 32 VALU", in, out, cyc);
    run<32, 1, 4>("8 MFMA x 4 VALU interleaved", in, out, cyc);
    run<64, 0, 0>("8 MFMA then 64 VALU", in, out, cyc);
    run<64, 1, 8>("8 MFMA x 8 VALU interleaved", in, out, cyc);
    run<64, 1, 6>("8 MFMA x 6 VALU + 16 tail", in, out, cyc);
    run<96, 0, 0>("8 MFMA then 96 VALU", in, out, cyc);
    run<48, 2, 0>("8 MFMA, 48 VALU on ITS result", in, out, cyc);
    run<56, 2, 0>("8 MFMA, 56 VALU on ITS result", in, out, cyc);
    run<72, 2, 0>("8 MFMA, 72 VALU on ITS result", in, out, cyc);
    run<80, 2, 0>("8 MFMA, 80 VALU on ITS result", in, out, cyc);
    run<88, 2, 0>("8 MFMA, 88 VALU on ITS result", in, out, cyc);
    run<112, 2, 0>("8 MFMA, 112 VALU on ITS result", in, out, cyc);
    run<128, 2, 0>("8 MFMA, 128 VALU on ITS result", in, out, cyc);
    run<32, 2, 0>("8 MFMA, 32 VALU on ITS result", in, out, cyc);
    run<64, 2, 0>("8 MFMA, 64 VALU on ITS result", in, out, cyc);
    run<96, 2, 0>("8 MFMA, 96 VALU on ITS result", in, out, cyc);
    run<96, 1, 12>("8 MFMA x 12 VALU interleaved", in, out, cyc);
    run<0, 3, 0>("16 MFMA 16x16x32 only", in, out, cyc);
    run<32, 3, 0>("16x16x32, 32 VALU on ITS result", in, out, cyc);
    run<48, 3, 0>("16x16x32, 48 VALU on ITS result", in, out, cyc);
    run<64, 3, 0>("16x16x32, 64 VALU on ITS result", in, out, cyc);
    run<72, 3, 0>("16x16x32, 72 VALU on ITS result", in, out, cyc);
    run<80, 3, 0>("16x16x32, 80 VALU on ITS result", in, out, cyc);
    run<96, 3, 0>("16x16x32, 96 VALU on ITS result", in, out, cyc);
    return 0;
}
