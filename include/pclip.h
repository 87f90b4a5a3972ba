/*
 * pclip.h — C ABI of the MI355X (gfx950) contrastive hot path: L2-normalise, CLIP BxB
 * logits + symmetric InfoNCE, FILIP token-wise max similarity + row InfoNCE, and the
 * backward of each.
 *
 * The reference (pan-emily/protein-clip) is pure Python and has no FFI of its own: the
 * "interface" each entry point replaces is the run of torch calls cited beside it
 * (paths relative to the reference checkout).  The Python host side
 * (protein-clip_amd/) binds these with ctypes and re-exports the reference's function
 * names on top.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous row-major memory owned by the caller
 *    (the one exception, the parameter pointer lists of pclip_head_tail_*, says so);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are
 *    enqueued on it and the call returns without synchronising;
 *  - nothing here allocates or frees device memory: scratch comes in through `ws`, sized
 *    by the matching *_workspace_bytes() query (256-byte aligned base expected);
 *  - return value 0 = enqueued; <0 = rejected before any launch, message from
 *    pclip_last_error() (thread-local);
 *  - scalars that live on the device in the reference (the learnable temperature, the
 *    upstream gradient of the loss) are passed as device pointers so no call ever forces a
 *    host sync.
 */
#ifndef PCLIP_H
#define PCLIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCLIP_OK 0
#define PCLIP_EINVAL (-1)   /* bad argument (null pointer, size out of the supported range) */
#define PCLIP_ELAUNCH (-2)  /* the HIP runtime refused a launch */
#define PCLIP_EDEVICE (-3)  /* not a gfx950 device / no device */

/* limits of this build (checked, never silently exceeded) */
#define PCLIP_CLIP_MAX_D 128     /* CLIP embedding width handled by the MFMA kernels      */
#define PCLIP_FILIP_MAX_D 128    /* FILIP token embedding width (zero-padded to 128)      */
#define PCLIP_FILIP_MAX_T 256    /* FILIP tokens per sequence handled by the fast kernels */
#define PCLIP_FILIP_MAX_T_LONG 4096 /* FILIP tokens per sequence at most (general kernel)  */
#define PCLIP_HEAD_MAX_BLOCKS 4  /* Linear-ReLU-LayerNorm-Dropout blocks of the embedding head */

int pclip_version(void);
const char* pclip_last_error(void);
/* 0 when device `dev` is a gfx950 part this library's code objects run on. */
int pclip_check_device(int dev);

/* ---------------------------------------------------------------------------------------
 * Row L2-normalise + temperature scale.
 * Replaces modules/models.py:50-51:  x = F.normalize(e, dim=-1) * exp(t / 2)
 *   e [B,d] f32, t device scalar, x [B,d] f32.   Any B >= 1, 1 <= d <= 4096.
 * ------------------------------------------------------------------------------------- */
int pclip_l2norm_scale_fwd(const float* e, const float* t, float* x, int B, int d, void* stream);

/* Backward of the above (autograd of models.py:50-51 under loss.backward(),
 * modules/training_utils.py:71).  dx is given as `nslab` partial slabs of [B,d] f32, slab
 * s at dx + s*slab_stride (elements), which are summed first (nslab = 1 for a plain dx).
 *   de [B,d] f32 is overwritten;  dt (device scalar) is ACCUMULATED into: dt += dL/dt,
 *   zero it first unless that is what you want (both towers add into one temperature).
 *   g: optional device scalar multiplying dx (NULL = 1). */
int pclip_l2norm_scale_bwd(const float* e, const float* t, const float* dx, int nslab, size_t slab_stride,
                           const float* g, float* de, float* dt, int B, int d, void* stream);

/* ---------------------------------------------------------------------------------------
 * CLIP symmetric InfoNCE.
 * Replaces modules/training_utils.py:85-89 (_contrastive_loss) and :260-264
 * (_contrastive_loss_gc): S = X Y^T; loss = 0.5*(CE_rows(S) + CE_cols(S)) with the diagonal
 * as target.  Computed with a max-shifted log-sum-exp (same value, no overflow).
 *
 * Sharded form: the caller owns rows [row_offset, row_offset+Br) of a global batch of Bg
 * pairs.  X_own/Y_own [Br,d] are its rows, X_all/Y_all [Bg,d] the gathered batch (for one
 * GPU: Br = Bg, row_offset = 0, *_own = *_all).
 *   lse_row [Br]  log sum_j exp(S[i,j])  for the caller's rows i
 *   lse_col [Br]  log sum_i exp(S[i,j])  for the caller's columns j
 *   loss_part     device scalar: (0.5/Bg) * sum_{own i} (lse_row_i + lse_col_i - 2 S_ii)
 *                 (= the loss itself on one GPU; summed over ranks otherwise)
 *   logits_out    optional [Br,Bg] f32: S (mode 1) or exp(S) (mode 2, what
 *                 get_logits returns, modules/training_utils.py:170-172); NULL/mode 0 = none
 * d % 4 == 0, d <= PCLIP_CLIP_MAX_D.
 * ------------------------------------------------------------------------------------- */
size_t pclip_clip_workspace_bytes(int Br, int Bg, int d);
int pclip_clip_fwd(const float* X_own, const float* Y_own, const float* X_all, const float* Y_all,
                   int Br, int Bg, int d, int row_offset,
                   float* lse_row, float* lse_col, float* loss_part,
                   float* logits_out, int logits_mode, void* ws, size_t ws_bytes, void* stream);

/* Backward (autograd of the same lines): with G = (0.5/Bg)(softmax_rows(S) + softmax_cols(S) - 2I),
 *   dX_own = g * G[own,:] Y_all,   dY_own = g * G[:,own]^T X_all.
 *   lse_row_all / lse_col_all [Bg]: the gathered vectors from pclip_clip_fwd.
 *   g: optional device scalar (upstream gradient of the loss; NULL = 1). */
int pclip_clip_bwd(const float* X_own, const float* Y_own, const float* X_all, const float* Y_all,
                   const float* lse_row_all, const float* lse_col_all,
                   int Br, int Bg, int d, int row_offset, const float* g,
                   float* dX_own, float* dY_own, void* ws, size_t ws_bytes, void* stream);

/* One-GPU fused step from raw head outputs (models.py:50-51 on both towers, then
 * training_utils.py:85-89, then loss.backward()): writes loss, d e_pep, d e_rec and dt
 * (dt is overwritten here).  Up to B = 512 the step is ONE persistent launch whose workgroups
 * meet at grid-wide barriers; that form is only taken when the CURRENT device keeps the whole grid
 * resident (a partition with few compute units takes the separate launches instead), so size the
 * workspace with the device current on which the step will run.
 * The one-launch form (also behind pclip_clip_loss_grad and pclip_clip_step_noalloc) needs its grid
 * resident as a whole, so two of them must not share a GPU: inside one process the library chains
 * such launches across streams (the new stream waits for the work enqueued on the stream used
 * last; streams under graph capture are left alone), ACROSS PROCESSES IT CANNOT — give the step a
 * GPU of its own (one process per GPU).  A launch whose barrier cannot complete gives up after
 * about half a second and fails visibly: loss, every gradient and dt are NaN, and so are those of
 * every later one-launch step on that GPU until pclip_clip_step_sync_reset(). */
size_t pclip_clip_step_workspace_bytes(int B, int d);
int pclip_clip_step(const float* e_pep, const float* e_rec, const float* t, int B, int d,
                    float* loss, float* de_pep, float* de_rec, float* dt,
                    void* ws, size_t ws_bytes, void* stream);

/* Loss and both gradients from the towers' OUTPUTS on one GPU (training_utils.py:85-89 followed by
 * loss.backward(), for x, y [B,d] as Encoder.forward returns them): loss (device scalar),
 * dx, dy [B,d] = d loss / d x, d loss / d y, all overwritten.  One launch up to B = 512 (see
 * pclip_clip_step), the forward and backward launches beyond. */
size_t pclip_clip_loss_grad_workspace_bytes(int B, int d);
int pclip_clip_loss_grad(const float* x, const float* y, int B, int d, float* loss, float* dx, float* dy,
                         void* ws, size_t ws_bytes, void* stream);

/* Diagnostics of the one-launch form's barrier (no counterpart in the reference).
 *   pclip_clip_step_sync_status: *gave_up = 1 when a barrier wait on the current device has given up
 *     since the last reset (results were NaN from then on).  SYNCHRONISES the device.
 *   pclip_clip_step_sync_reset: clears the barrier words of the current device (enqueued on `stream`);
 *     call it only while no one-launch step is in flight on that GPU.
 *   pclip_clip_step_selftest_giveup(on): while on, every one-launch step waits for one arrival more
 *     than its grid has, i.e. takes the give-up path (tests of the visible failure). */
int pclip_clip_step_sync_status(int* gave_up);
int pclip_clip_step_sync_reset(void* stream);
int pclip_clip_step_selftest_giveup(int on);

/* The same step without a workspace argument.  It has no counterpart in the reference: it exists
 * for callers that cannot allocate device memory or size a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  The scratch of the step lives in
 * a fixed device array inside the library (pclip_clip_step_noalloc_max_bytes() = 10 MiB per GPU,
 * held until the process ends), which covers B <= 1024 at d = 128; larger problems are rejected
 * with PCLIP_EINVAL.  No call allocates or frees.  The first call on a device looks the array's
 * address up (once per device, with that device current) and asks the runtime for the kernel's
 * occupancy: make one eager call per device before capturing the step into a HIP graph, where it
 * then is a single kernel node (B <= 512).  All calls on one GPU share that GPU's
 * array: calls that overlap in time on one GPU (two streams, two threads) race; calls on one stream
 * are safe.  Results equal pclip_clip_step's. */
size_t pclip_clip_step_noalloc_max_bytes(void);
int pclip_clip_step_noalloc(const float* e_pep, const float* e_rec, const float* t, int B, int d,
                            float* loss, float* de_pep, float* de_rec, float* dt, void* stream);

/* ---------------------------------------------------------------------------------------
 * Row-wise InfoNCE with the diagonal as target.
 * Replaces modules/training_utils_filip.py:15-23 (_contrastive_loss, incl. use_dcl).
 *   logits [Br,Bc] f32; the target of row i is column i + row_offset.
 *   lse [Br] out;  loss_part: device scalar (1/Bg) * sum_i (lse_i - logits[i, i+off]).
 * Backward: dlogits[i,j] = g * (1/Bg) * (softmax_i(j) - [j == i+off])   (dcl: the diagonal is
 * absent from the softmax).
 * ------------------------------------------------------------------------------------- */
int pclip_rowce_fwd(const float* logits, int Br, int Bc, int row_offset, int Bg, int use_dcl,
                    float* lse, float* loss_part, void* stream);
int pclip_rowce_bwd(const float* logits, const float* lse, int Br, int Bc, int row_offset, int Bg,
                    int use_dcl, const float* g, float gscale, float* dlogits, void* stream);

/* Both FILIP losses with their gradients in two launches.  Replaces
 * modules/training_utils_filip.py:26-39 (_compute_loss = (rowCE(sim_A) + rowCE(sim_B)) / 2) and
 * the backward of that expression; equal to two pclip_rowce_fwd plus two pclip_rowce_bwd
 * (use_dcl = 0, gscale = 0.5) followed by the average.
 *   sim_A, sim_B [Br,Bc] f32;  lse2 [2,Br] out;  g_A, g_B [Br,Bc] out: d loss / d sim
 *   loss3 [3] out: {(lossA + lossB) / 2, lossA, lossB}, each the share of these Br rows (1/Bg) sum
 *   ws_terms: 2*Br floats of scratch */
int pclip_filip_loss(const float* sim_A, const float* sim_B, int Br, int Bc, int row_offset, int Bg,
                     float* lse2, float* loss3, float* g_A, float* g_B, float* ws_terms, void* stream);

/* ---------------------------------------------------------------------------------------
 * Masked mean over tokens.  Replaces Encoder._masked_mean (modules/models.py:76-90), the
 * pooling in front of the embedding head of each CLIP tower (models.py:48):
 *   mean[b,:] = sum_t h[b,t,:] * mask[b,t] / max(sum_t mask[b,t], 1)
 * h [B,T,d] f32, mask [B,T] u8 (used by value, as the reference multiplies by the mask),
 * mean [B,d] f32, inv_count [B] f32 = 1 / max(sum mask, 1) (kept for the backward).
 * Backward: dh[b,t,:] = mask[b,t] * g[b,:] * inv_count[b]   (dh is OVERWRITTEN).
 * When d is a multiple of 4, h, g and dh must be 16-byte aligned (rejected otherwise).
 * ------------------------------------------------------------------------------------- */
int pclip_masked_mean_fwd(const float* h, const uint8_t* mask, int B, int T, int d, float* mean, float* inv_count, void* stream);
int pclip_masked_mean_bwd(const float* g, const uint8_t* mask, const float* inv_count, int B, int T, int d, float* dh, void* stream);

/* ---------------------------------------------------------------------------------------
 * Embedding head of a CLIP tower, from the pooled vector to the scaled unit embedding.
 * Replaces modules/models.py:49-51 with embedding_ffn as built at :54-73 (h2 - 1 blocks of
 * Linear -> ReLU -> LayerNorm -> Dropout, then one Linear):
 *   x_0 = pooled;  a_l = relu(W_l x_l + b_l);  x_{l+1} = (LayerNorm(a_l) * gamma_l + beta_l) * keep_l * keep_scale;
 *   z = W_nblk x_nblk + b_nblk;  out = z / max(|z|, 1e-12) * exp(t / 2).
 * pooled, out [B,d] f32; d % 4 == 0, d <= 128; nblk <= PCLIP_HEAD_MAX_BLOCKS.
 * params: a HOST array of 4*nblk + 2 device pointers in module order: per block W [d,d] (row-major
 *   [out][in], as nn.Linear holds it), b [d], gamma [d], beta [d]; then the last W and b.
 * keep: the dropout masks [nblk,B,d] u8 (nonzero = kept) drawn by the caller, keep_scale = 1/(1-p);
 *   NULL = no dropout (eval mode).  The reference draws them inside nn.Dropout from torch's
 *   generator; which elements are dropped is the caller's random stream here, the arithmetic is the same.
 * ln_eps: the LayerNorm epsilon (1e-5 in the reference's modules).
 * save: pclip_head_tail_save_floats(B, d, nblk) floats kept for the backward (activations, row
 *   statistics, z).  t: the temperature as a device scalar.
 * Backward (autograd of the same lines): dout [B,d] -> dpooled [B,d] and dparams (host array of
 * device pointers shaped and ordered like params), all OVERWRITTEN; dt is ACCUMULATED
 * (dt += 0.5 * sum(out * dout)).  Two launches: rows, then parameter gradients (fixed summation
 * order: results are reproducible except dt, which is met with float atomics).
 * ------------------------------------------------------------------------------------- */
size_t pclip_head_tail_save_floats(int B, int d, int nblk);
size_t pclip_head_tail_bwd_workspace_bytes(int B, int d, int nblk);
int pclip_head_tail_fwd(const float* pooled, const float* const* params, int nblk, const uint8_t* keep, float keep_scale,
                        float ln_eps, const float* t, int B, int d, float* save, float* out, void* stream);
int pclip_head_tail_bwd(const float* pooled, const float* const* params, int nblk, const uint8_t* keep, float keep_scale,
                        const float* t, const float* save, const float* dout, int B, int d,
                        float* dpooled, float* const* dparams, float* dt, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * FILIP token-wise similarity.
 * Replaces modules/models.py:238-239 (token normalise, no epsilon) and :292-356
 * (_filip_similarity_score): sim[i,j,a,b] = hA[i,a].hB[j,b] / t; padded pairs -> finfo.min;
 * sim_A[i,j] = mean over valid a of max_b, sim_B[i,j] = mean over valid b of max_a.
 * The contraction runs on bf16 matrix cores with f32 accumulation.
 *
 * Step 1, prep: h [B,T,d] f32 -> hn [B,Tp,128] bf16 (Tp = T rounded up to 32; pad rows and
 * pad columns zero), optionally normalised per token, and negated when tsign (optional
 * device scalar, NULL = never) is negative: pass the temperature for the A side, so that
 * max(x/t) = max(-x)/|t| also holds for t < 0 without a host read of t; maskw [B,Tp/32]
 * u32 = the mask bits of each 32-token tile (bit r = token 32*tile + r; any nonzero mask
 * byte counts as valid, as .bool() does at models.py:323-324).
 * ------------------------------------------------------------------------------------- */
int pclip_filip_prep(const float* h, const uint8_t* mask, int B, int T, int d, int normalise, const float* tsign,
                     uint16_t* hn, uint32_t* maskw, void* stream);

/* Step 2, scores.  hnA [Br,TAp,128], hnB [Bc,TBp,128] bf16 from prep; `t` is the temperature
 * as a device scalar (the kernel uses 1/|t|; the sign was folded into hnA by prep).
 * unit_norm != 0 asserts that every token row has L2 norm <= 1 (prep with normalise = 1):
 * the kernel then keeps value and arg-max in one integer maximum per element (exact up to the
 * f32 rounding of an accumulation that starts at 4.0: maxima within 2e-6, near-ties closer than
 * that may pick either token); unit_norm = 0 makes no assumption (exact f32 maxima, slower).
 * A value of 2 or more is unit_norm = 1 with that number of workgroups aimed at per launch
 * instead of the default of 768 (tuning and tests; results do not depend on it).
 *   sim_A, sim_B [Br,Bc] f32
 *   argA [Br,Bc,TAp] u16: for each valid a the b attaining the max
 *   argB [Br,Bc,TBp] u16: for each valid b the a attaining the max
 * (argA/argB feed the backward, which reads the entries of valid tokens only; entries of
 * padding tokens are unspecified and may be left unwritten.  Pass NULL for both to skip them
 * in inference.)
 * TA, TB <= PCLIP_FILIP_MAX_T_LONG; sequences beyond PCLIP_FILIP_MAX_T tokens (the reference keeps
 * chains of up to 2,000 residues, modules/data_utils_2protein.py:88) run on a general, untuned
 * kernel here; pclip_filip_sim_fwd_ws below runs them at the fast kernel's rate. */
int pclip_filip_sim_fwd(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA,
                        const uint16_t* hnB, const uint32_t* maskwB, int Bc, int TB,
                        const float* t, int unit_norm, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB,
                        void* stream);

/* The same for a BLOCK OF COLUMNS of wider outputs: sim_A, sim_B are [Br,Bc_total], argA [Br,Bc_total,TAp],
 * argB [Br,Bc_total,TBp], and this call fills their columns col_offset .. col_offset + Bc - 1 from the Bc
 * sequences given (every score depends on its own sequence pair only, models.py:332-351, so any split
 * into column blocks gives the same bits).  It lets a rank of a multi-GPU job score its own B
 * sequences while the all-gather of the other ranks' is still in flight.  pclip_filip_sim_fwd is
 * this call with col_offset = 0, Bc_total = Bc. */
int pclip_filip_sim_fwd_cols(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA,
                             const uint16_t* hnB, const uint32_t* maskwB, int Bc, int TB,
                             const float* t, int unit_norm, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB,
                             int col_offset, int Bc_total, void* stream);

/* The same with a workspace, which sequences beyond PCLIP_FILIP_MAX_T tokens with unit-norm operands
 * (unit_norm != 0) use to run at the fast kernel's rate: every sequence is taken as segments of 256
 * tokens, the streaming kernel scores all segment pairs and leaves the maximum and its partner of
 * every token per segment pair in `ws`, and a second kernel joins the segments of each sequence pair
 * (on equal maxima the token of the later segment).  pclip_filip_sim_fwd_workspace_bytes gives the size
 * (0 when no workspace is used; it is bounded — the columns are worked off in blocks — at about 256 MB
 * plus one A-side block).  With ws = NULL this is pclip_filip_sim_fwd_cols. */
size_t pclip_filip_sim_fwd_workspace_bytes(int Br, int TA, int Bc, int TB, int unit_norm);
int pclip_filip_sim_fwd_ws(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA,
                           const uint16_t* hnB, const uint32_t* maskwB, int Bc, int TB,
                           const float* t, int unit_norm, float* sim_A, float* sim_B, uint16_t* argA, uint16_t* argB,
                           int col_offset, int Bc_total, void* ws, size_t ws_bytes, void* stream);

/* Step 3, backward of the scores w.r.t. the bf16 operands (autograd of models.py:332-351:
 * the max routes each gradient to its arg-max partner).  gA, gB [Br,Bc] f32 = d loss / d sim_A,
 * d loss / d sim_B.  Outputs (f32, OVERWRITTEN): dnA [Br,TA,128p] and dnB [Bc,TB,128p] hold
 * d loss / d (normalised operand) in the padded [.,Tp,128] layout of prep; dt is
 * ACCUMULATED: dt += -(1/t) * sum(gA*sim_A + gB*sim_B).  `ws` holds the bucketed token lists
 * (sized by pclip_filip_bwd_workspace_bytes; about 4*Br*Bc*(TAp+TBp) bytes, plus four
 * partial copies of dnA and dnB when the operand images exceed the L2 window budget of 4 MB).
 * The order in
 * which a row's contributions are summed is not fixed, so results may differ in the last
 * bits from run to run.  Either of dnA, dnB may be NULL: that side is skipped (dt is produced
 * together with dnA), so a caller can run the B side first and start exchanging dnB between
 * GPUs while the A side is computed.  The token images of one side may not exceed 4 GB
 * (Br*TAp and Bc*TBp at most 16,777,215 rows: 32-bit row offsets). */
size_t pclip_filip_bwd_workspace_bytes(int Br, int TA, int Bc, int TB);
int pclip_filip_sim_bwd(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA,
                        const uint16_t* hnB, const uint32_t* maskwB, int Bc, int TB,
                        const float* t, const float* sim_A, const float* sim_B,
                        const uint16_t* argA, const uint16_t* argB,
                        const float* gA, const float* gB,
                        float* dnA, float* dnB, float* dt, void* ws, size_t ws_bytes, void* stream);
/* The second of two one-sided calls on the same workspace (same inputs, the other side): the
 * marks of the token pairs that chose each other, which the first call left in `ws`, are used
 * again instead of being recomputed.  No counterpart in the reference; it serves the split
 * described above (modules/models.py:221-245 has one backward for both sides). */
int pclip_filip_sim_bwd_resume(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA,
                               const uint16_t* hnB, const uint32_t* maskwB, int Bc, int TB,
                               const float* t, const float* sim_A, const float* sim_B,
                               const uint16_t* argA, const uint16_t* argB,
                               const float* gA, const float* gB,
                               float* dnA, float* dnB, float* dt, void* ws, size_t ws_bytes, void* stream);

/* Steps 3 and 4 in one call, for callers that want d loss / d (raw token embeddings) and have
 * no use for dn (no gradient exchange between the two): same inputs as pclip_filip_sim_bwd plus
 * the raw embeddings hA [Br,TA,d], hB [Bc,TB,d] that went into pclip_filip_prep (with this
 * `normalise`, and with `t` as the A side's tsign); outputs dhA [Br,TA,d], dhB [Bc,TB,d]
 * (OVERWRITTEN), dt accumulated.  The window partials are summed inside the prep backward, so
 * dn is never written.  Workspace: pclip_filip_bwd_fused_workspace_bytes. */
size_t pclip_filip_bwd_fused_workspace_bytes(int Br, int TA, int Bc, int TB);
int pclip_filip_bwd_fused(const uint16_t* hnA, const uint32_t* maskwA, int Br, int TA,
                          const uint16_t* hnB, const uint32_t* maskwB, int Bc, int TB,
                          const float* t, const float* sim_A, const float* sim_B,
                          const uint16_t* argA, const uint16_t* argB, const float* gA, const float* gB,
                          const float* hA, const float* hB, int d, int normalise,
                          float* dhA, float* dhB, float* dt, void* ws, size_t ws_bytes, void* stream);

/* Step 4, backward of prep: dn [B,Tp,128] f32 (padded layout) -> dh [B,T,d] f32 through the
 * normalise Jacobian (I - n n^T)/|h| when normalise = 1 (identity otherwise), sign flipped
 * when tsign (as in prep) is negative.  When `accumulate` is nonzero dh += instead of dh =. */
int pclip_filip_prep_bwd(const float* h, const float* dn, int B, int T, int d, int normalise, const float* tsign,
                         int accumulate, float* dh, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCLIP_H */
