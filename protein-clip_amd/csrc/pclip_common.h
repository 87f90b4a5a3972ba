// Shared host/device helpers for the pclip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>
#include "../../include/pclip.h"

#ifndef FLT_MAX
#define FLT_MAX 3.402823466e+38f
#endif

// ---- host side -------------------------------------------------------------------------
extern thread_local char g_pclip_err[512];

inline int pclip_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_pclip_err, sizeof(g_pclip_err), fmt, ap);
    va_end(ap);
    return code;
}

#define PCLIP_REQUIRE(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) return pclip_fail(PCLIP_EINVAL, __VA_ARGS__); \
    } while (0)

inline int pclip_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pclip_fail(PCLIP_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return PCLIP_OK;
}

inline size_t pclip_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Dynamic-LDS launch limit of one kernel.  The attribute belongs to the (kernel, device) pair, so
// what has been raised is remembered per device; the lock makes concurrent host threads safe.
// Nothing here influences a result: it only decides whether hipFuncSetAttribute is called again.
#define PCLIP_MAX_DEVICES 64
struct PclipLdsLimit {
    std::mutex mu;
    int raised[PCLIP_MAX_DEVICES] = {};
};
inline int pclip_ensure_lds(PclipLdsLimit& lim, const void* fn, size_t bytes, const char* who) {
    if (bytes <= 48 * 1024) return PCLIP_OK;   // within the default limit of every kernel
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return pclip_fail(PCLIP_EDEVICE, "%s: no current device", who);
    std::lock_guard<std::mutex> lock(lim.mu);
    const bool known = dev >= 0 && dev < PCLIP_MAX_DEVICES;
    if (known && lim.raised[dev] >= (int)bytes) return PCLIP_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return pclip_fail(PCLIP_ELAUNCH, "%s: cannot raise the dynamic LDS limit to %zu bytes", who, bytes);
    if (known) lim.raised[dev] = (int)bytes;
    return PCLIP_OK;
}

// ---- device side -----------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Row of a 32x32 MFMA accumulator tile held in register `reg` (0..15) by lane half `h` (0/1).
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Sum over the 64 lanes, result in every lane: four DPP steps inside each row of 16 lanes (no
// LDS round trip), then two cross-row exchanges.
__device__ __forceinline__ float wave_sum(float v) {
#define PCLIP_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
    PCLIP_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    PCLIP_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    PCLIP_DPP_ADD(0x141);  // row_half_mirror
    PCLIP_DPP_ADD(0x140);  // row_mirror
#undef PCLIP_DPP_ADD
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// internal launcher of the streaming unit-norm FILIP score kernel (filip_stream.hip); wgs_target:
// workgroups to aim at per launch (<= 0: the default)
int filip_sim_stream_launch(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TAp, const uint16_t* hnB, const uint32_t* maskwB, int Bc,
                            int TBp, const float* t, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB, int col_offset, int ldc,
                            int wgs_target, void* stream);
// the same with the sequences given as segments of 256 tokens (see FilipStreamArgs) and the per-token maxima as extra outputs
int filip_sim_stream_launch_seg(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TAp, int segA, int rowsA, const uint16_t* hnB,
                                const uint32_t* maskwB, int Bc, int TBp, int segB, int tilesB, const float* t, float* sim_A, float* sim_B,
                                uint16_t* argA, uint16_t* argB, float* rowval, float* colval, int col_offset, int ldc, int wgs_target,
                                void* stream);
// internal launcher of the general FILIP score kernel for long sequences (filip_long.hip)
int filip_sim_long_launch(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TAp, const uint16_t* hnB, const uint32_t* maskwB, int Bc,
                          int TBp, const float* t, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB, int col_offset, int ldc,
                          void* stream);
// internal launchers of the two-tower normalise kernels used by the fused CLIP step (l2norm.hip)
int l2norm_scale_fwd_pair_launch(const float* e0, const float* e1, const float* t, float* x0, float* x1, int B, int d, float* zero_me,
                                 void* stream);
int l2norm_scale_bwd_pair_launch(const float* e0, const float* e1, const float* t, const float* dx0, const float* dx1, int nslab,
                                 size_t slab_stride, float* de0, float* de1, float* dt, int B, int d, void* stream);
// internal launcher of the prep backward over a sum of gradient slabs (filip.hip)
int filip_prep_bwd_launch(const float* h, const float* dn, int nslab, size_t slab_stride, size_t row_stride, const uint32_t* mw, int B, int T, int d,
                          int normalise, const float* tsign, int accumulate, float* dh, void* stream);

// long unit-norm sequences as 256-token segments on the streaming kernel, joined by a second kernel (filip_long.hip)
size_t filip_sim_long_stream_ws_bytes(int Br, int TAp, int Bc, int TBp);
int filip_sim_long_stream_launch(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TAp, const uint16_t* hnB, const uint32_t* maskwB, int Bc,
                                 int TBp, const float* t, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB, int col_offset, int ldc,
                                 void* ws, void* stream);

// ---- the one-launch CLIP step for B <= 512 (clip_step.hip) ----
bool clip_step_fused_ok(int B, int d);
size_t clip_step_fused_ws_bytes(int B, int d);
int clip_step_fused_launch(const float* e_pep, const float* e_rec, const float* t, int B, int d, float* loss, float* de_pep, float* de_rec,
                           float* dt, void* ws, void* stream);
