// Argument blocks shared by the CLIP loss kernels (clip.hip, clip_tiled.hip).
#pragma once
#include "pclip_common.h"

struct ClipProb {
    const float* A;     // [Br,d]  the caller's rows
    const float* Bm;    // [Bc,d]  every row of the other side
    int Br, Bc, off;    // the target column of row i is i + off
    float2* part;       // fwd: [npart][Br] running (max, sum) per column split and wave
    float* diag;        // fwd: [Br] S[i, i+off]
    float* lse_out;     // fwd: [Br]
    float* logits;      // fwd: optional [Br,Bc]
    const float* lseA;  // bwd: [Br] log-sum-exp of the caller's rows
    const float* lseB;  // bwd: [Bc] log-sum-exp of every column in the other direction
    float* slab;        // bwd: [nsplit][Br][d] partial dA
};

struct ClipArgs {
    ClipProb p[2];
    int d;
    int nsplit;
    int npart;       // partial (max, sum) pairs per row: nsplit x 4 (one per wave) or nsplit (tiled kernels)
    int tiles_per_split;
    int logits_mode;
    float scale;     // 0.5 / global batch
    const float* g;  // optional upstream gradient (device scalar)
    unsigned* fin_counter;  // fwd: ticket of the finalize blocks (cleared by the score kernel in front of them)
    float* fin_part;        // fwd: [finalize blocks] partial loss sums
};

// the tiled kernels for tall row blocks (clip_tiled.hip): four waves = four 32-row tiles of A against one staged tile of Bm
#define CLIP_TILED_MIN_ROWS 128
size_t clip_tiled_lds_bytes(bool backward);
int clip_lse_tiled_launch(const ClipArgs& a, int Br, hipStream_t st);
int clip_grad_tiled_launch(const ClipArgs& a, int Br, hipStream_t st);
