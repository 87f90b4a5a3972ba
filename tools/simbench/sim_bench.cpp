// Standalone timing harness for the FILIP score kernels (no torch): random unit-norm-bounded
// bf16 images at the benchmark shape, N timed launches with HIP events.
//   hipcc -O3 --offload-arch=gfx950 [-DABL_...] -I../../include sim_bench.cpp ../../protein-clip_amd/csrc/filip_stream.hip ../../protein-clip_amd/csrc/filip_fast.hip -o sim_bench
//   ./sim_bench [reps] [full|ragged] [stream|old] [arg|noarg]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

thread_local char g_pclip_err[512];
int filip_sim_unit_launch(const uint16_t*, const uint32_t*, int, int, const uint16_t*, const uint32_t*, int, int, const float*, float*, float*,
                          uint16_t*, uint16_t*, void*);
int filip_sim_stream_launch(const uint16_t*, const uint32_t*, int, int, const uint16_t*, const uint32_t*, int, int, const float*, float*, float*,
                            uint16_t*, uint16_t*, void*);

#ifdef STREAM_STAMPS
extern unsigned long long* g_stream_dbg;
#endif
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    const bool full = argc > 2 && !strcmp(argv[2], "full");
    const bool old = argc > 3 && !strcmp(argv[3], "old");
    const bool noarg = argc > 4 && !strcmp(argv[4], "noarg");
    const int B = argc > 5 ? atoi(argv[5]) : 256;
    const int Bc = argc > 6 ? atoi(argv[6]) : B;
    const int L = 256, D = 128;
    srand(1);
    // "a<len>": every A sequence has <len> valid tokens, B sequences are full length
    const int fixedA = (argc > 2 && argv[2][0] == 'a') ? atoi(argv[2] + 1) : 0;
    auto fill = [&](std::vector<uint16_t>& img, std::vector<uint32_t>& mw, int n, double& toks, int fixed) {
        img.assign((size_t)n * L * D, 0); mw.assign((size_t)n * 8, 0); toks = 0;
        for (int s = 0; s < n; ++s) {
            int len = fixed ? fixed : full ? L : 128 + rand() % 129;
            toks += len;
            for (int t = 0; t < len; ++t) {
                mw[s * 8 + t / 32] |= 1u << (t % 32);
                for (int k = 0; k < D; ++k) img[((size_t)s * L + t) * D + k] = f2bf(((rand() % 2001) - 1000) * 0.00008f);
            }
        }
    };
    std::vector<uint16_t> hA, hB; std::vector<uint32_t> mA, mB; double tokA, tokB;
    fill(hA, mA, B, tokA, fixedA); fill(hB, mB, Bc, tokB, fixedA ? L : 0);
    uint16_t *dA, *dB, *argA, *argB; uint32_t *dmA, *dmB; float *t, *sA, *sB;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dmA, mA.size() * 4); hipMalloc(&dmB, mB.size() * 4);
    hipMalloc(&argA, (size_t)B * Bc * L * 2); hipMalloc(&argB, (size_t)B * Bc * L * 2);
    hipMalloc(&t, 4); hipMalloc(&sA, (size_t)B * Bc * 4); hipMalloc(&sB, (size_t)B * Bc * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dmA, mA.data(), mA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dmB, mB.data(), mB.size() * 4, hipMemcpyHostToDevice);
    float one = 1.f; hipMemcpy(t, &one, 4, hipMemcpyHostToDevice);
#ifdef STREAM_STAMPS
    const size_t ndbg = (size_t)8192 * 16 * 8;
    hipMalloc(&g_stream_dbg, ndbg * 8); hipMemset(g_stream_dbg, 0, ndbg * 8);
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int r = 0; r < reps + 3; ++r) {
        hipEventRecord(e0, 0);
        int rc = old ? filip_sim_unit_launch(dA, dmA, B, L, dB, dmB, Bc, L, t, sA, sB, noarg ? nullptr : argA, noarg ? nullptr : argB, 0)
                     : filip_sim_stream_launch(dA, dmA, B, L, dB, dmB, Bc, L, t, sA, sB, noarg ? nullptr : argA, noarg ? nullptr : argB, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (rc) { printf("launch failed: %s\n", g_pclip_err); return 1; }
        float m; hipEventElapsedTime(&m, e0, e1);
        if (r >= 3) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<float> hs((size_t)B * Bc); hipMemcpy(hs.data(), sA, hs.size() * 4, hipMemcpyDeviceToHost);
    double chk = 0; for (float v : hs) chk += v;
#ifdef STREAM_STAMPS
    {   // per-segment shader-clock cycles of the last launch (stream kernel only)
        std::vector<unsigned long long> hd(ndbg); hipMemcpy(hd.data(), g_stream_dbg, ndbg * 8, hipMemcpyDeviceToHost);
        const char* nm[8] = {"dma-issue", "meta", "read+mfma", "keys+atomic", "row-epilogue", "vmcnt+lgkm", "barrier", "finalize+emit"};
        double sum[8] = {0}, tot = 0; size_t nw = 0;
        for (size_t g = 0; g < 8192; ++g) for (int w = 0; w < 16; ++w) {
            double t = 0; for (int k = 0; k < 8; ++k) t += hd[(g * 16 + w) * 8 + k];
            if (t == 0) continue;
            ++nw; for (int k = 0; k < 8; ++k) sum[k] += hd[(g * 16 + w) * 8 + k];
        }
        for (int k = 0; k < 8; ++k) tot += sum[k];
        for (int g : {0, 517}) for (int w = 0; w < 16; ++w) { printf("  wg %d wave %2d:", g, w); for (int k = 0; k < 8; ++k) printf(" %8llu", hd[((size_t)g * 16 + w) * 8 + k]); printf("\n"); }
        for (int k = 0; k < 8; ++k) printf("  seg %-14s %6.1f%%  %.0f cycles/wave\n", nm[k], 100 * sum[k] / tot, sum[k] / nw);
    }
#endif
    const double flops = 2.0 * D * tokA * tokB;
    printf("%s %s %s B=%d Bc=%d: median %.3f ms  min %.3f ms  -> %.1f TFLOP/s (median)  checksum %.6f\n", old ? "old" : "stream", full ? "full" : "ragged",
           noarg ? "noarg" : "arg", B, Bc, ms[ms.size() / 2], ms[0], flops / (ms[ms.size() / 2] * 1e-3) / 1e12, chk / hs.size());
    return 0;
}
