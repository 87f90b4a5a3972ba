// Microbenchmark: two ways for one wave (two A tiles, B tile stream in LDS) to overlap matrix chains with key work, every
// instruction pinned (volatile asm between scheduling barriers), tags in scalar registers rebuilt before every region.
//   V1: region A = chain of tile t, A0 || keys of (t-1, A1);  region B = chain of (t, A1) || keys of (t, A0)   [one dependent
//       chain at a time per wave]
//   V2: one region per tile: the chains of (t, A0) and (t, A1) alternate (two independent chains), between them the keys of
//       both products of tile t-1 (a second pair of accumulators)
//   hipcc -O3 --offload-arch=gfx950 dual_chain_model.cpp -o dual_chain_model && ./dual_chain_model
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ unsigned long long shader_cycles() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__device__ __forceinline__ unsigned long long ticks_100mhz() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define SET_TAGS(bt4) _Pragma("unroll") for (int r = 0; r < 16; ++r) asm volatile("s_or_b32 %0, %1, %2" : "=s"(tag[r]) : "s"(bt4), "n"(r) : "scc")
#define KEYS_STEP(ACC, KM, CB, s)                                                                                  \
    do {                                                                                                           \
        int ka_, kb_;                                                                                              \
        asm volatile("v_lshl_or_b32 %0, %1, 7, %2" : "=v"(ka_) : "v"(ACC[2 * (s)]), "s"(tag[2 * (s)]));              \
        asm volatile("v_lshl_or_b32 %0, %1, 7, %2" : "=v"(kb_) : "v"(ACC[2 * (s) + 1]), "s"(tag[2 * (s) + 1]));      \
        asm volatile("v_max_i32 %0, %0, %1" : "+v"(KM[2 * (s)]) : "v"(ka_));                                        \
        asm volatile("v_max_i32 %0, %0, %1" : "+v"(KM[2 * (s) + 1]) : "v"(kb_));                                    \
        asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(CB) : "v"(ka_), "v"(kb_));                                  \
    } while (0)

template <int NW, int V, int RELOADS>
__global__ __launch_bounds__(NW * 64) void k4(const bf16x8* __restrict__ in, int* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long long* col = reinterpret_cast<long long*>(smem + 65536);
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8 a0[8], a1[8], b[8];
    for (int s = 0; s < 8; ++s) { a0[s] = in[(tid & 255) * 16 + s]; a1[s] = in[(tid & 255) * 16 + 8 + s]; }
    for (int i = tid; i < 65536 / 16; i += NW * 64) reinterpret_cast<bf16x8*>(smem)[i] = in[i & 4095];
    for (int i = tid; i < 512; i += NW * 64) col[i] = 0;
    __syncthreads();
    int km0[16], km1[16], tag[16];
    for (int r = 0; r < 16; ++r) { km0[r] = -r; km1[r] = r; tag[r] = 0; }
    f32x16 four;
    for (int r = 0; r < 16; ++r) { four[r] = 4.f; if (V < 3) asm volatile("" : "+v"(four[r])); }
    f32x16 x0 = four, x1 = four, y0 = four, y1 = four;
    typedef __attribute__((address_space(3))) const bf16x8 lds_frag;
    { lds_frag* cur = (lds_frag*)(smem + lane * 16);
#pragma unroll
      for (int s = 0; s < 8; ++s) b[s] = cur[64 * s]; }
    const unsigned long long c0_ = shader_cycles(), r0_ = ticks_100mhz();
    for (int it = 0; it < iters; it += 2) {
        if (V == 1) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                lds_frag* same = (lds_frag*)(smem + ((it + u) & 7) * 8192 + lane * 16);
                lds_frag* next = (lds_frag*)(smem + ((it + u + 1) & 7) * 8192 + lane * 16);
                { SET_TAGS(((it + u) & 7) << 4);
                  int cb = 0x80000000; SB();
#pragma unroll
                  for (int s = 0; s < 8; ++s) { x0 = MFMA(a0[s], b[s], s == 0 ? four : x0); SB(); if (RELOADS > 1) b[s] = same[64 * s]; KEYS_STEP(x1, km1, cb, s); SB(); }
                  atomicMax(&col[(((it + u) & 7) * 2 + 1) * 32 + (lane & 31)], ((long long)cb << 32) | lane); }
                { SET_TAGS(((it + u) & 7) << 4);
                  int cb = 0x80000000; SB();
#pragma unroll
                  for (int s = 0; s < 8; ++s) { x1 = MFMA(a1[s], b[s], s == 0 ? four : x1); SB(); b[s] = next[64 * s]; KEYS_STEP(x0, km0, cb, s); SB(); }
                  atomicMax(&col[(((it + u) & 7) * 2) * 32 + (lane & 31)], ((long long)cb << 32) | lane); }
            }
        } else if (V == 7) {
            // matrix chains only, two A tiles per wave: every fragment set feeds two chains
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                lds_frag* next = (lds_frag*)(smem + ((it + u + 1) & 7) * 8192 + lane * 16);
#pragma unroll
                for (int s = 0; s < 8; ++s) { x0 = MFMA(a0[s], b[s], s == 0 ? four : x0); x1 = MFMA(a1[s], b[s], s == 0 ? four : x1); b[s] = next[64 * s]; }
                asm volatile("" : "+v"(x0), "+v"(x1));
                km0[u] ^= __float_as_int(x0[u]) ^ __float_as_int(x1[u]);
            }
        } else if (V == 5 || V == 6) {
            // matrix chains only (V5: operands stay in registers; V6: fragments re-read from LDS behind their use)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                lds_frag* next = (lds_frag*)(smem + ((it + u + 1) & 7) * 8192 + lane * 16);
#pragma unroll
                for (int s = 0; s < 8; ++s) { x0 = MFMA(a0[s], b[s], s == 0 ? four : x0); if (V == 6) b[s] = next[64 * s]; }
                asm volatile("" : "+v"(x0));
                km0[u] ^= __float_as_int(x0[u]);
            }
        } else if (V == 3 || V == 4) {
            // one A tile per wave.  V3: chain of tile t + 1 (one accumulator) between the keys of tile t (the other).  V4: plain order
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                lds_frag* next = (lds_frag*)(smem + ((it + u + 1) & 7) * 8192 + lane * 16);
                SET_TAGS(((it + u) & 7) << 4);
                int cb = 0x80000000;
                if (V == 3) {
                    SB();
                    if (u == 0) {
#pragma unroll
                        for (int s = 0; s < 8; ++s) { x0 = MFMA(a0[s], b[s], s == 0 ? four : x0); SB(); b[s] = next[64 * s]; KEYS_STEP(x1, km0, cb, s); SB(); }
                    } else {
#pragma unroll
                        for (int s = 0; s < 8; ++s) { x1 = MFMA(a0[s], b[s], s == 0 ? four : x1); SB(); b[s] = next[64 * s]; KEYS_STEP(x0, km0, cb, s); SB(); }
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 8; ++s) { x0 = MFMA(a0[s], b[s], s == 0 ? four : x0); b[s] = next[64 * s]; }
#pragma unroll
                    for (int s = 0; s < 8; ++s) KEYS_STEP(x0, km0, cb, s);
                }
                atomicMax(&col[(((it + u) & 7) * 2) * 32 + (lane & 31)], ((long long)cb << 32) | lane);
            }
        } else {
            // tile it: chains into x0, x1; keys of y0, y1.  tile it + 1: chains into y0, y1; keys of x0, x1
#define TILE(C0, C1, K0, K1, t)                                                                                 \
            { lds_frag* next = (lds_frag*)(smem + (((t) + 1) & 7) * 8192 + lane * 16);                         \
              SET_TAGS(((t) & 7) << 4);                                                                        \
              int cb0 = 0x80000000, cb1 = 0x80000000; SB();                                                    \
              _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                  \
                  C0 = MFMA(a0[s], b[s], s == 0 ? four : C0); SB();                                            \
                  KEYS_STEP(K0, km0, cb0, s); SB();                                                            \
                  C1 = MFMA(a1[s], b[s], s == 0 ? four : C1); SB();                                            \
                  b[s] = next[64 * s];                                                                         \
                  KEYS_STEP(K1, km1, cb1, s); SB();                                                            \
              }                                                                                                \
              atomicMax(&col[(((t) & 7) * 2) * 32 + (lane & 31)], ((long long)cb0 << 32) | lane);              \
              atomicMax(&col[(((t) & 7) * 2 + 1) * 32 + (lane & 31)], ((long long)cb1 << 32) | lane); }
            TILE(x0, x1, y0, y1, it)
            TILE(y0, y1, x0, x1, it + 1)
        }
        if ((it & 3) == 2) __builtin_amdgcn_s_barrier();
    }
    int x = 0;
    for (int r = 0; r < 16; ++r) x ^= km0[r] ^ __float_as_int(x0[r]) ^ __float_as_int(x1[r]);
    if (V < 3) for (int r = 0; r < 16; ++r) x ^= km1[r] ^ __float_as_int(y0[r]) ^ __float_as_int(y1[r]);
    out[blockIdx.x * NW * 64 + tid] = x;
    if (blockIdx.x == 0 && tid == 0) {   // shader-clock cycles and 100-MHz ticks of the loop: the clock the SIMDs really ran at
        unsigned long long* cl = reinterpret_cast<unsigned long long*>(out + 256 * 1024 + 8);
        cl[0] = shader_cycles() - c0_;
        cl[1] = ticks_100mhz() - r0_;
    }
}

template <int NW, int V, int RELOADS>
static void run(const char* name, bf16x8* in, int* out) {
    const int iters = 4000;
    (void)hipFuncSetAttribute((const void*)k4<NW, V, RELOADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k4<NW, V, RELOADS>), dim3(256), dim3(NW * 64), 65536 + 4096, 0, in, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / iters / (NW / 4) / ((V >= 3 && V != 7) ? 1 : 2);   // per product and SIMD
    unsigned long long cl[2] = {0, 0};
    (void)hipMemcpy(cl, out + 256 * 1024 + 8, sizeof(cl), hipMemcpyDeviceToHost);
    const double mhz = cl[1] ? (double)cl[0] / ((double)cl[1] / 100.0) : 0.0;
    printf("%-64s %6.1f ns per product and SIMD  (%4.0f TFLOP/s)  shader clock %4.0f MHz\n", name, ns, 1024.0 * 8 * 32768 / ns * 1e-3, mhz);
}

int main() {
    bf16x8* in; int* out;
    (void)hipMalloc(&in, 4096 * sizeof(bf16x8)); (void)hipMemset(in, 0, 4096 * sizeof(bf16x8));
    (void)hipMalloc(&out, 256 * 1024 * 4 + 64);
    run<8, 1, 1>("8 waves, one chain at a time, fragments read once", in, out);
    run<8, 1, 2>("8 waves, one chain at a time, fragments read twice", in, out);
    run<8, 2, 1>("8 waves, two chains alternating || keys of the tile before", in, out);
    run<4, 1, 1>("4 waves (one per SIMD), one chain at a time", in, out);
    run<4, 2, 1>("4 waves (one per SIMD), two chains alternating", in, out);
    run<16, 4, 1>("16 waves x 1 tile, plain order", in, out);
    run<12, 4, 1>("12 waves x 1 tile, plain order", in, out);
    run<8, 4, 1>("8 waves x 1 tile, plain order", in, out);
    run<16, 3, 1>("16 waves x 1 tile, chain of the next tile || keys", in, out);
    run<12, 3, 1>("12 waves x 1 tile, chain of the next tile || keys", in, out);
    run<8, 3, 1>("8 waves x 1 tile, chain of the next tile || keys", in, out);
    run<16, 5, 1>("16 waves, matrix chains only, registers, zero operands", in, out);
    run<16, 6, 1>("16 waves, matrix chains only, LDS fragments, zero operands", in, out);
    run<8, 7, 1>("8 waves x 2 tiles, matrix chains only, zero operands", in, out);
    run<16, 7, 1>("16 waves x 2 tiles, matrix chains only, zero operands", in, out);
    // the same with random operands (bf16 in [-0.125, 0.125]): matrix power and clocks depend on the data
    {
        unsigned short* h = new unsigned short[4096 * 8];
        unsigned x = 12345u;
        for (int i = 0; i < 4096 * 8; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x80ffu) | 0x3d00u); }
        (void)hipMemcpy(in, h, 4096 * sizeof(bf16x8), hipMemcpyHostToDevice);
        run<16, 4, 1>("16 waves x 1 tile, plain order, RANDOM operands", in, out);
        run<12, 4, 1>("12 waves x 1 tile, plain order, RANDOM operands", in, out);
        run<12, 3, 1>("12 waves x 1 tile, chain of the next tile || keys, RANDOM operands", in, out);
        run<8, 4, 1>("8 waves x 1 tile, plain order, RANDOM operands", in, out);
        run<8, 1, 1>("8 waves, one chain at a time, RANDOM operands", in, out);
        run<16, 5, 1>("16 waves, matrix chains only, registers, RANDOM operands", in, out);
        run<16, 6, 1>("16 waves, matrix chains only, LDS fragments, RANDOM operands", in, out);
        run<8, 7, 1>("8 waves x 2 tiles, matrix chains only, RANDOM operands", in, out);
        run<12, 7, 1>("12 waves x 2 tiles, matrix chains only, RANDOM operands", in, out);
        run<16, 7, 1>("16 waves x 2 tiles, matrix chains only, RANDOM operands", in, out);
    }
    return 0;
}
