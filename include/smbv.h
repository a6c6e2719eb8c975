/* smbv.h — C ABI of libsmbv.so: the MI355X (gfx950) hot path of the smb-vision 3-D masked-image-modeling train step.
 *
 * Drop-in boundary: the reference has no native code and no FFI; its hot path is reached through the
 * Python model-class interface called by HF Trainer (reference src/run_mim.py:345-357, 429-436).  This library is what
 * the Python mirror of that interface (smb_vision_amd/modeling_videomae.py) binds with ctypes; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.  Each entry point names the reference code it replaces (paths relative
 * to the reference repo root; "spec" = src/models/videomae/modeling_videomae.py).
 *
 * Conventions for EVERY function:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *   - the caller owns every buffer, including workspaces; the library never allocates or frees device memory,
 *     never synchronises, never changes the current device; work is enqueued on `stream` (a hipStream_t);
 *   - bf16 tensors are uint16 storage; matrices are row-major with an explicit leading dimension where one is taken;
 *   - pointers must be 16-byte aligned unless noted;
 *   - return value: 0 = ok, negative = SMBV_ERR_* (bad shape / alignment / argument / workspace), positive = hipError_t;
 *     smbv_last_error() returns the message of the calling thread's last failure; no C++ exception crosses the ABI;
 *   - stream-ordered; callable from any host thread (autograd's backward thread included), but the step / block / attention-backward entry
 *     points fork onto the per-device side streams below and are NOT safe to run from two host threads on the SAME device at once (the library
 *     serialises EVERY record / wait on its per-device events with a per-device mutex - round 5: the deferred weight-gradient launches' completion
 *     events included - so the runtime calls never overlap on one event and a re-recorded event only makes a wait cover a later point of the in-order side
 *     stream; the side streams themselves are shared, i.e. two threads' steps serialise their side work and that is the extent of the support);
 *   - state inside the library: the immutable code object; a thread-local error string; per device, two private non-blocking side streams and a
 *     dozen events created on first use and kept for the life of the process (a block's weight-gradient products run as one grouped launch on a side
 *     stream beside the NEXT block's backward, the step forwards their weight casts beside the first kernels; everything forked is joined back into
 *     `stream` before the call returns); a 64-KiB `__device__` array of scheduling words for the attention backward (block roles and tile queues of
 *     attn_bwd_pair_kernel - never results): eight slots, each OWNED by one stream of the device (the first eight streams that run an attention
 *     backward; a ninth takes a two-launch path with no shared state), every word stamped with its launch's epoch, so launches that share a slot
 *     are always ordered by one stream and a word of another launch is detected, not consumed; 4 bytes of host-mapped status per device
 *     (smbv_device_status below); and the kernel-selection word of smbv_debug_set_variant() below (0 unless a tuning tool sets it).  Nothing else
 *     persists between calls.  The step / attention entry points are not meant to be captured into a hipGraph (the launch epochs are host counters).
 */
#ifndef SMBV_H
#define SMBV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMBV_OK 0
#define SMBV_ERR_SHAPE (-1)
#define SMBV_ERR_ALIGN (-2)
#define SMBV_ERR_ARG (-3)
#define SMBV_ERR_WORKSPACE (-4)
#define SMBV_ERR_STATE (-5)      /* a kernel of an earlier call recorded a fault on this device: see smbv_device_status */
#define SMBV_MAX_LAYERS 48

typedef void* smbv_stream;   /* hipStream_t */
typedef uint16_t smbv_bf16;

/* ABI version = major*1000 + minor.  A binding must check major == SMBV_ABI_MAJOR (struct layouts / argument meanings are only stable within one
 * major) and minor >= the newest entry point it calls.  2000: every *_cfg struct opens with `struct_bytes` (= sizeof of the caller's struct, checked
 * by the library: a binding compiled against a shorter or longer layout is refused with SMBV_ERR_ARG instead of being read past its end);
 * 1000-1002 (round 1-2) had no such guard and grew smbv_mim_cfg / smbv_vjepa_cfg in place.  3000 (round 4): smbv_attention_fwd / smbv_attention_bwd take
 * the residual output O_lo, the *_cfg structs end with attn_o_resid.  3001 (round 5, additive): smbv_gemm_nt_splitk + its workspace query; 3002 (round 5, additive): smbv_attention_bwd_split + its workspace query;
 * 3003 (round 5, additive): smbv_gemm_tn_grouped_split + its workspace query; 3004 (round 5, additive): smbv_rope3d_cfg, smbv_gemm_nt_rope, smbv_attention_bwd_rope.
 * 3005 (round 6, additive): smbv_debug_pair_state(0) also restores the hand-off flags / tickets of the split launches and takes op 2 + 4 e; device status codes 0x5004 / 0x5005.
 * 3006 (round 6, additive): smbv_encoder_hidden_states / smbv_mim_hidden_states (output_hidden_states=True of the reference's model classes). */
#define SMBV_ABI_MAJOR 3
int smbv_version(void);
int smbv_last_error(char* buf, size_t len);               /* copies the thread's last error text; returns its length */
/* Tuning / A-B hook used by tools/ and by the kernel-equivalence tests, NOT by the product path: selects alternative kernels for the same
 * arithmetic (process-wide, not thread-safe; 0 = the shipped selection).  Bits: 2 attention backward query-owner blocks on the 32x32x16 MFMA (shipped: 16x16x32); 32 register-staged GEMMs; 256 first-generation 128x128 NT
 * GEMM everywhere; 4 the blocks' weight-gradient products inline on the side stream (one launch each + finishing launches) instead of deferred and grouped; 131072 second-generation ring NT GEMM everywhere (default: by shape); 512 four-slot ring; 262144 ring with 64-deep K slices; 1024 no mixed-tile NT launches (all tiles 128 x 128); 2048 attention forward with the guarded sweep only (no optimistic first sweep); 4096 delta = rowsum(dO * O) by its own launch instead of the
 * out-projection input-gradient epilogue; 8192 attention dQ and dK/dV on
 * one stream (with 524288); 524288 attention backward as two launches (dK/dV, dQ) instead of the one mixed-role launch; 16384 weight- and input-gradient GEMMs on one stream; 32768 / 65536 TN tile selection; bits 20-23 GEMM ablation (timing only); 1<<24 attention forward on the plain (tile, head, batch) grid instead of the
 * XCD-aware block order; 1<<25 attention backward kernels on the XCD-aware block order; bits 26-28 start-up stagger
 * units of the attention kernels (forward default 2, backward 0), 1<<29 no stagger; 1<<30 the library's side streams at the lowest priority (read once, when
 * they are created). */
void smbv_debug_set_variant(int bits);
/* (A second A/B word, for tools only, is read ONCE from the environment variable SMBV_DEBUG_AB when the library is loaded - csrc/capi.hip lists its bits; unset
 * / 0 on the product path.) */
/* The device's sticky status (ABI 3000).  A kernel that detects a fault it cannot return - the attention backward's block scheduler finding a
 * scheduling word of another launch (0x5001 / 0x5003) or both of its tile queues exhausted (0x5002: a tile was skipped), a consumer half of a split
 * launch that outwaited its producer (0x5004: attention split sweeps, split grouped weight gradients, split-K product), a block of the split
 * weight-gradient launch that drew a ticket outside its launch's run (0x5005) - stores a code into 4 bytes of host-mapped memory; every
 * smbv_*_forward / _backward / smbv_attention_bwd call on that device then returns SMBV_ERR_STATE instead of enqueueing more work on top of incomplete
 * gradients.  This function reads it without synchronising (a fault shows once its kernel has run): 0, or SMBV_ERR_STATE with the code in
 * smbv_last_error(); clear != 0 resets the WORD.  The scheduling state a faulted launch left behind (tile queues, tickets, hand-off flags) is put back
 * by smbv_debug_pair_state(0, stream) - recover with both, in that order, after the stream has drained.  Hand-off flags are stamped with their launch's
 * epoch / sequence number, so even without the reset a word of a faulted launch is never taken for a later launch's.  Never set on a healthy run. */
int smbv_device_status(int clear);
/* Test / recovery hook for the above: op 0 restores every scheduling slot of the current device - attention backward tile queues, per-CU tickets and
 * split-sweep flags, the split weight-gradient launch's tickets and flags - and the host's launch epochs to their load-time state (enqueued on `stream`; the
 * device must be idle otherwise).  NOT the product path: op 1 dirties the slots the way a second launch sharing a slot would (attention key-owner queue
 * + 3, weight-gradient ticket + 3); op 2 + 4 e leaves every hand-off flag raised to e + 1: what an unfinished attention launch of epoch e / an unfinished
 * weight-gradient launch of sequence number e + 1 would have left behind. */
int smbv_debug_pair_state(int op, smbv_stream stream);

/* ---- per-op entry points (also what the parity tests call) ------------------------------------------------------ */

/* Epilogue of smbv_gemm_nt (all optional; see csrc/smbv_kernels.h GemmEpi). */
typedef struct smbv_gemm_epi {
  float alpha;
  const float* bias;
  const float* rowadd; const int32_t* rowidx; int ldr;
  const smbv_bf16* dgelu_pre; int ld_pre;
  int act_gelu; smbv_bf16* aux_pre; int ld_aux;
  smbv_bf16* out_bf16; int ld_ob;
  float* out_f32; int ld_of;
  int grp_in, grp_out;
} smbv_gemm_epi;

/* C[M,N] = epi(A[M,K] · B[N,K]^T): every nn.Linear forward of the path (spec :261-264, :311-313, :368-370, :384-386,
 * :801-803, :720-722) with B = the HF weight [out,in]; and their input gradients with B = a transposed bf16 copy. */
int smbv_gemm_nt(const smbv_bf16* A, int lda, const smbv_bf16* B, int ldb, int M, int N, int K, const smbv_gemm_epi* epi,
                 smbv_stream stream);
/* The same product as 256 x 128 output tiles with the K range as TWO halves per tile (ABI 3001; built for the long-K / narrow-N Linears - fc2 forward,
 * spec :384-386, and the input gradients of fc1 and of the fused q | k | v projection - measured SLOWER there than smbv_gemm_nt's own selection and
 * therefore NOT what the steps run, profiles/r05_splitk_ab.txt; kept as a tested op): the halves are computed by two
 * workgroups, the first publishing its fp32 partial tile through `ws`, the second adding it and running the epilogue (two addends: run-to-run
 * identical; the result differs from smbv_gemm_nt's in the last fp32 bit of the K-sum, nothing else).  ws: smbv_gemm_nt_splitk_workspace_bytes(M, N),
 * 256-byte aligned, contents arbitrary (the call clears its flag words on `stream`).  SMBV_ERR_SHAPE when the product is not one this form is built for
 * (N multiple of 128, K multiple of 64 and >= 1152, 96..256 tiles, a plain epilogue: no GELU / dGELU). */
int smbv_gemm_nt_splitk(const smbv_bf16* A, int lda, const smbv_bf16* B, int ldb, int M, int N, int K, const smbv_gemm_epi* epi, void* ws, size_t ws_bytes,
                        smbv_stream stream);
size_t smbv_gemm_nt_splitk_workspace_bytes(int M, int N);
/* out[N1,N2] (fp32) = sum_m X[m,N1] · Y[m,N2]: the weight gradients of the same Linears (autograd of the above). */
int smbv_gemm_tn(const smbv_bf16* X, int ldx, const smbv_bf16* Y, int ldy, int M, int N1, int N2, float* out, void* ws,
                 size_t ws_bytes, smbv_stream stream);
size_t smbv_gemm_tn_workspace_bytes(int M, int N1, int N2);
/* the same product plus, from the same pass over X, its column sums xsum[N1] = sum_m X[m, :] - weight and bias gradient of one nn.Linear
 * (dW = dY^T A, db = sum over tokens of dY) in one launch; ws as above */
int smbv_gemm_tn_bias(const smbv_bf16* X, int ldx, const smbv_bf16* Y, int ldy, int M, int N1, int N2, float* out, float* xsum, void* ws,
                      size_t ws_bytes, smbv_stream stream);

/* Several such weight (+ bias) gradients as ONE launch (ABI 3000): every output tile is reduced over its product's whole token range - no slabs, no
 * workspace, no finishing launch; xsum segments as in smbv_gemm_tn_bias: column sum n of X goes to sum_dst[n / seg][n % seg] (NULL segment: dropped;
 * seg = 0: no sums).  1..8 jobs; every N1, N2 a multiple of 128 (256 x 256 tiles when all are multiples of 256).  The block backward issues a
 * transformer block's four products this way, deferred to the end of the block and run beside the next block's backward. */
typedef struct smbv_tn_job {
  const smbv_bf16* X; const smbv_bf16* Y; float* out;
  float* sum_dst[3];
  int ldx, ldy, M, N1, N2, seg;
  int reserved;                   /* set by the library */
} smbv_tn_job;
int smbv_gemm_tn_grouped(const smbv_tn_job* jobs /* host */, int njobs, smbv_stream stream);
/* The same launch with every tile's token range run as two halves by two workgroups (ABI 3003): for a launch that has the chip to itself and fewer tiles than
 * half its CUs - the steps run the LAST block's weight gradients this way, the optimizer waits for nothing else (profiles/r05_tn_split_ab.txt).  The
 * first half's workgroup hands its fp32 tile and column sums through ws (write-through stores, one flag per tile), the second adds its own and writes
 * the result: the same two addends in the same order on every run.  Every N1, N2 a multiple of 256, every M >= 256, at most 128 tiles in all
 * (the workspace query returns 0 otherwise and the call SMBV_ERR_SHAPE); ws 256-byte aligned and not shared between launches that may overlap.  The
 * launch's scheduling words (a unit ticket, one flag per tile) live in a slot of the library's that belongs to `stream` - eight slots per device, a slot taken by
 * the first launch of a stream; launches of a ninth stream run whole (same products, the two addends of a tile then meet in the matrix pipe instead).  Units are
 * claimed from the ticket, so a consumer half only ever runs after every producer half has been claimed by a running workgroup; flags carry the launch's
 * sequence number (ABI 3005: round 5 kept ONE flag array per device and took roles from the workgroup index). */
size_t smbv_gemm_tn_grouped_split_workspace_bytes(const smbv_tn_job* jobs /* host */, int njobs);
int smbv_gemm_tn_grouped_split(const smbv_tn_job* jobs /* host */, int njobs, void* ws, size_t ws_bytes, smbv_stream stream);

/* nn.LayerNorm over the last dim (spec :402-403, :412, :423, :676/:721); y is the bf16 GEMM operand.
 * Input rows are read at (row / grp_in) * grp_out + row % grp_in + row_off when grp_in > 0. */
int smbv_layernorm_fwd(const float* x, const float* gamma, const float* beta, smbv_bf16* y, float* mean, float* rstd, int rows,
                       int D, float eps, long grp_in, long grp_out, long row_off, smbv_stream stream);
int smbv_layernorm_bwd(const smbv_bf16* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* dres, float* dx, smbv_bf16* dx_bf16, float* dgamma, float* dbeta, int rows, int D, long grp_in,
                       long grp_out, long row_off, void* ws, size_t ws_bytes, smbv_stream stream);
size_t smbv_layernorm_bwd_workspace_bytes(int rows, int D);

/* softmax_fp32(Q K^T * scale) V, Dh = 64 (spec :196-223, dispatched at :270-289) or Dh = 32 (the V-JEPA predictor's heads, src/models/vjepa/
 * modeling_vjepa.py:251-253); any other Dh is SMBV_ERR_SHAPE (the step functions run such heads zero-padded in 64-wide slots).  Q,K,V: column
 * slices of one token-major [B*S, ld] matrix (head h at columns h*Dh..); O: [B*S, ldo]; LSE/delta: fp32 [B,H,Sq].
 * Sq/q_off: only the queries q_off .. q_off+Sq-1 of every sequence are computed (keys/values: all S); Sq <= 0 means all.
 * (The last decoder block only needs its masked tokens' outputs: spec :717-718 keeps hidden_states[:, -return_token_num:].)
 * O_lo (ABI 3000; may be NULL): same layout as O; the forward stores bf16(O_fp32 - O) there and the backward, given the same pointer, forms
 * delta = rowsum(dO * (O + O_lo)), i.e. from the output to 16 significant bits.  A bf16 O alone puts a 2^-9 relative error into delta that the softmax
 * Jacobian multiplies by the probability-weighted MEAN key / query - a term that does not average out over many near-uniform keys. */
int smbv_attention_fwd(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, smbv_bf16* O, smbv_bf16* O_lo, int ldo, float* LSE,
                       int B, int S, int H, int Dh, float scale, int Sq, int q_off, smbv_stream stream);
/* The same forward with OCP e4m3 operands on the block-scaled fp8 MFMA (BASELINE.json config 5: "fp8 MFMA QK^T / PV"): Q (pre-scaled), K per
 * (token, head, 32 channels) and V per (channel, 32 keys) under power-of-two E8M0 scales, P in e4m3 under a fixed 2^-7 scale; fp32 softmax and
 * accumulation.  ws: smbv_attention_fp8_workspace_bytes(B, S, H) bytes, 256-B aligned (the quantised operands).  Feeds the bf16 backward above. */
size_t smbv_attention_fp8_workspace_bytes(int B, int S, int H);
int smbv_attention_fwd_fp8(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, smbv_bf16* O, int ldo, float* LSE, int B, int S, int H,
                           int Dh, float scale, int Sq, int q_off, void* ws, size_t ws_bytes, smbv_stream stream);
/* smbv_attention_bwd as the steps run it since round 5 (ABI 3002): given scratch, the one mixed-role launch runs the LAST tiles of each role - 192 of each when the
 * role has at least twice as many - as TWO half sweeps by two workgroups (a key-owner tile over the first / second half of the query tiles, a query-owner tile over the
 * first / second half of the key tiles), the first publishing its fp32 accumulators through ws, the second adding them and storing the gradients: the launch's
 * partial last round of whole ~170-us tiles becomes half as coarse (-0.33 ms per ViT-base MIM step).  Two addends: run-to-run identical; against smbv_attention_bwd the
 * split tiles' sums associate differently in fp32, before the one bf16 rounding.  ws: smbv_attention_bwd_split_workspace_bytes(B, S, H, Dh) bytes (0 when the shape is
 * too small for any split), 256-B aligned, contents arbitrary; ws = NULL: exactly smbv_attention_bwd.  fp8_ws (may be NULL): as smbv_attention_bwd_fp8's ws - the query-owner blocks then
 * recompute S and dP on the fp8 MFMA, half sweeps included. */
size_t smbv_attention_bwd_split_workspace_bytes(int B, int S, int H, int Dh);
int smbv_attention_bwd_split(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, const smbv_bf16* O, const smbv_bf16* O_lo, const smbv_bf16* dO,
                             int ldo, const float* LSE, float* delta, smbv_bf16* dQ, smbv_bf16* dK, smbv_bf16* dV, int ldd, int B, int S, int H, int Dh,
                             float scale, int parts, int Sq, int q_off, void* ws, size_t ws_bytes, void* fp8_ws, size_t fp8_ws_bytes, smbv_stream stream);
/* smbv_attention_bwd_split for an attention whose q and k went through the rotary map `rope` (smbv_rope3d_cfg below the V-JEPA kernels; ABI 3004): dQ and dK - the
 * q | k sections of one [B * S, ldd] gradient matrix - leave through the TRANSPOSED map, applied to the bf16-rounded gradients in the kernel's stores: bitwise
 * what smbv_attention_bwd_split followed by smbv_rope3d(dQ, ldd, dK - dQ, ..., transpose = 1) leaves (the fp8 blocks and the debug forms run exactly those two).
 * rope->head_stride = Dh (the attention's head slot), rope->Dh = the real head_dim (<= the slot), parts = 6 or 7; ws / fp8_ws as smbv_attention_bwd_split. */
struct smbv_rope3d_cfg;
int smbv_attention_bwd_rope(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, const smbv_bf16* O, const smbv_bf16* O_lo, const smbv_bf16* dO,
                            int ldo, const float* LSE, float* delta, smbv_bf16* dQ, smbv_bf16* dK, smbv_bf16* dV, int ldd, int B, int S, int H, int Dh,
                            float scale, int parts, int Sq, int q_off, void* ws, size_t ws_bytes, void* fp8_ws, size_t fp8_ws_bytes,
                            const struct smbv_rope3d_cfg* rope, smbv_stream stream);
/* smbv_attention_bwd with the query-owner blocks' two recomputed products (S, dP) on the block-scaled fp8 MFMA (head_dim 64, parts = 6 or 7): Q, K, V, dO are
 * quantised to e4m3 rows with E8M0 scales per (token, head, 32 channels) into ws (smbv_attention_bwd_fp8_workspace_bytes, 256-B aligned) by four small
 * launches in front of the pair launch. */
size_t smbv_attention_bwd_fp8_workspace_bytes(int B, int S, int H);
int smbv_attention_bwd_fp8(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, const smbv_bf16* O, const smbv_bf16* O_lo, const smbv_bf16* dO,
                           int ldo, const float* LSE, float* delta, smbv_bf16* dQ, smbv_bf16* dK, smbv_bf16* dV, int ldd, int B, int S, int H, int Dh, float scale,
                           int parts, int Sq, int q_off, void* ws, size_t ws_bytes, smbv_stream stream);
int smbv_attention_bwd(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, const smbv_bf16* O, const smbv_bf16* O_lo, const smbv_bf16* dO,
                       int ldo, const float* LSE, float* delta, smbv_bf16* dQ, smbv_bf16* dK, smbv_bf16* dV, int ldd, int B, int S,
                       int H, int Dh, float scale, int parts /* bit0 delta, bit1 dK/dV, bit2 dQ; 7 = all */, int Sq, int q_off, smbv_stream stream);

/* Input feed, device side: the per-volume tail of the reference's MONAI pipeline - ScaleIntensityRanged ->
 * SpatialPadd -> CenterSpatialCropd -> PermuteImage (src/dataloader/mim.py:154-169, 88-93) - in one pass from the uploaded raw volume
 * to the model input.  src: [B, H0, W0, D0] (D0 fastest; what EnsureChannelFirstd leaves of a NIfTI array), src_dtype 0 int16,
 * 1 uint16, 2 bf16, 3 fp32; dst: fp32 [B, D, 1, H, W].  v = (x - a_min) / (a_max - a_min) * (b_max - b_min) + b_min, clipped to
 * [b_min, b_max] when clip != 0; symmetric zero pad, centre crop. */
enum { SMBV_SRC_I16 = 0, SMBV_SRC_U16 = 1, SMBV_SRC_BF16 = 2, SMBV_SRC_F32 = 3 };
int smbv_prepare_volume(const void* src, int src_dtype, float* dst, int B, int H0, int W0, int D0, int H, int W, int D, float a_min,
                        float a_max, float b_min, float b_max, int clip, smbv_stream stream);

/* Brick readers over pixel_values fp32 [B,T,1,H,W] (spec :172-177,190-191 Conv3d patchify; :835-867 labels; :896-897 MSE).
 * idx: int32 token ids (0..N-1) of the selected bricks, `per_batch` per volume, rows = B*per_batch. */
int smbv_gather_bricks(const float* pix, const int32_t* idx, int rows, int per_batch, smbv_bf16* out, int T, int H, int W, int ts,
                       int ps, smbv_stream stream);
int smbv_brick_stats(const float* pix, const int32_t* idx, int rows, int per_batch, float* mean, float* rinv, int T, int H, int W,
                     int ts, int ps, smbv_stream stream);
int smbv_mse_fwd(const float* pix, const int32_t* idx, int rows, int per_batch, const float* mean, const float* rinv,
                 const smbv_bf16* logits, float* row_partial, float* loss, int T, int H, int W, int ts, int ps, smbv_stream stream);
int smbv_mse_bwd(const float* pix, const int32_t* idx, int rows, int per_batch, const float* mean, const float* rinv,
                 const smbv_bf16* logits, smbv_bf16* dlogits, const float* gscale, int T, int H, int W, int ts, int ps,
                 smbv_stream stream);

int smbv_cast_f32_bf16(const float* src, smbv_bf16* dst, size_t n, smbv_stream stream);
int smbv_transpose_cast_f32_bf16(const float* src, smbv_bf16* dst, int R, int C, int ldd, smbv_stream stream);
int smbv_colsum_bf16(const smbv_bf16* X, int ld, int M, int N, float* out, void* ws, size_t ws_bytes, smbv_stream stream);
size_t smbv_colsum_workspace_bytes(int M, int N);

/* ---- the whole MIM train step: VideoMAEForPreTraining.forward (spec :753-908) and its autograd -------------------- */

typedef struct smbv_mim_cfg {
  int struct_bytes;               /* sizeof(smbv_mim_cfg) as the caller compiled it */
  int B, T, H, W;                 /* volume batch: pixel_values fp32 [B,T,1,H,W] */
  int ps, ts;                     /* patch_size, tubelet_size */
  int N, Nvis, Nmask;             /* tokens per volume: all / visible / masked */
  int d, heads, ff, layers;       /* encoder: hidden_size, num_attention_heads, intermediate_size, num_hidden_layers.  Head widths d / heads (and
                                     dd / dheads): multiples of 8 up to 64; 64 and 32 have native attention builds, the others run zero-padded in
                                     64-wide slots (exact; SMBV_ERR_SHAPE above 64; the fp8 path: 64 only) */
  int dd, dheads, dff, dlayers;   /* decoder_* */
  float ln_eps;                   /* config.layer_norm_eps (blocks) */
  float dec_norm_eps;             /* decoder.norm eps: torch default 1e-5 (spec :676) */
  int recompute;                  /* 1 = gradient checkpointing (spec :459-465, torch.utils.checkpoint per layer): keep only every block's input, share
                                     two activation sets per stack (even / odd blocks: a block's deferred weight-gradient launch reads one while the next
                                     block re-runs into the other) and re-run a block's forward in front of its backward.  Same results bit for bit,
                                     one extra forward of time, ~2/layers of the activation memory */
  int attn_fp8;                   /* BASELINE config 5, "fp8 MFMA QK^T / PV": bit 0 = attention forward with OCP e4m3 operands on the block-scaled fp8 MFMA;
                                     bit 1 (ABI 3000) = also the backward's recomputed S = Q K^T and dP = dO V^T (its query-owner blocks; the accumulating
                                     dV / dK / dQ products stay bf16); 0 = bf16 throughout.  3 is what config.smbv_attention_fp8 = True selects */
  int attn_o_resid;               /* 1 = every block's attention forward also saves the bf16 residual of its fp32 output, O_lo = bf16(O_fp32 - bf16(O_fp32)), and the
                                     backward forms delta = rowsum(dO * (O + O_lo)) (see smbv_attention_fwd).  One more [rows, width] bf16 activation per block;
                                     ignored with attn_fp8 */
} smbv_mim_cfg;

/* fp32 tensors of one VideoMAELayer in the pinned-era (vendored) state-dict layout. */
typedef struct smbv_layer_ptrs {
  float *ln1_w, *ln1_b;           /* layernorm_before */
  float *q_w, *k_w, *v_w;         /* attention.attention.{query,key,value}.weight [w,w] */
  float *q_b, *v_b;               /* attention.attention.{q_bias,v_bias} [w] (key has no bias) */
  float *proj_w, *proj_b;         /* attention.output.dense */
  float *ln2_w, *ln2_b;           /* layernorm_after */
  float *fc1_w, *fc1_b;           /* intermediate.dense [ff,w] */
  float *fc2_w, *fc2_b;           /* output.dense [w,ff] */
} smbv_layer_ptrs;

/* One instance holds the parameters, a second one (same shape) the gradients. */
typedef struct smbv_mim_tensors {
  float *pe_w, *pe_b;             /* videomae.embeddings.patch_embeddings.projection [d,1,ts,ps,ps], [d] */
  smbv_layer_ptrs enc[SMBV_MAX_LAYERS];
  float *e2d_w;                   /* encoder_to_decoder.weight [dd,d] */
  float *mask_token;              /* [1,1,dd] */
  smbv_layer_ptrs dec[SMBV_MAX_LAYERS];
  float *dnorm_w, *dnorm_b;       /* decoder.norm */
  float *head_w, *head_b;         /* decoder.head [K,dd], [K] */
} smbv_mim_tensors;

typedef struct smbv_mim_io {
  const float* pixel_values;      /* fp32 [B,T,1,H,W] */
  const int32_t* vis_idx;         /* int32 [B,Nvis]  ascending token ids with bool_masked_pos == 0 */
  const int32_t* mask_idx;        /* int32 [B,Nmask] ascending token ids with bool_masked_pos == 1 */
  const float* pos_enc;           /* fp32 [N,d]  sin-cos table (spec :95-106), resident on device */
  const float* pos_dec;           /* fp32 [N,dd] */
  float* loss;                    /* fp32 [1] */
  smbv_bf16* logits;              /* bf16 [B*Nmask, K] */
  const float* grad_scale;        /* backward only: fp32 [1] upstream d(loss) (NULL = 1.0) */
} smbv_mim_io;

/* bytes of the activation/weight-copy/scratch workspace one step needs (fwd writes it, bwd reads it). */
size_t smbv_mim_workspace_bytes(const smbv_mim_cfg* cfg);
/* host pointers to cfg / params / io structs; device pointers inside them. */
int smbv_mim_forward(const smbv_mim_cfg* cfg, const smbv_mim_tensors* params, const smbv_mim_io* io, void* ws, size_t ws_bytes,
                     smbv_stream stream);
/* Gradients of loss*grad_scale w.r.t. every tensor in `grads` (overwritten, not accumulated).  `events`/`n_events`:
 * optional hipEvent_t array recorded as gradient groups complete - a block's event on the library's side stream, behind the block's deferred
 * weight-gradient launch; the step's own groups on `stream`, or through the side stream while launches are pending there: the events COMPLETE in array
 * order, so a consumer may gate a range of groups on the last one's event - in this order:
 *   [0] head+decoder.norm, [1..dlayers] decoder layers (last first), [dlayers+1] mask_token+encoder_to_decoder,
 *   [..+layers] encoder layers (last first), [last] patch embedding  — used to launch RCCL all-reduces on a side stream. */
int smbv_mim_backward(const smbv_mim_cfg* cfg, const smbv_mim_tensors* params, const smbv_mim_tensors* grads, const smbv_mim_io* io,
                      void* ws, size_t ws_bytes, void** events, int n_events, smbv_stream stream);
/* output_hidden_states=True of VideoMAEForPreTraining (spec :900-907 hands on the ENCODER's hidden states, visible tokens only): after smbv_mim_forward,
 * out[l] (l = 0 .. layers; n_out = layers + 1) points at the fp32 [B*Nvis, d] residual stream in front of encoder layer l (out[layers] = the encoder's output;
 * MIM has no final encoder LayerNorm, spec :517-520).  Kept by every mode (gradient checkpointing keeps exactly these).  Pointers into `ws`, no launch. */
int smbv_mim_hidden_states(const smbv_mim_cfg* cfg, void* ws, size_t ws_bytes, float** out, int n_out);

/* ---- encoder-only forward: VideoMAEModel.forward (spec :509-655) as src/run_inference.py:78-83 calls it for the
 *      embedding export (all tokens, no mask) and as the classification head (spec :917-989) sits on it. ---- */

typedef struct smbv_encoder_cfg {
  int struct_bytes;               /* sizeof(smbv_encoder_cfg) as the caller compiled it */
  int B, T, H, W;                 /* pixel_values fp32 [B,T,1,H,W] */
  int ps, ts;
  int N, Nsel;                    /* tokens per volume; tokens processed (N without a mask, the visible ones with bool_masked_pos) */
  int d, heads, ff, layers;
  float ln_eps;                   /* config.layer_norm_eps */
  int final_ln;                   /* VideoMAEModel.layernorm: present iff !config.use_mean_pooling (spec :517-520) */
  float final_ln_eps;
  int train;                      /* 0: inference (one shared activation set); 1: every layer keeps its activations for smbv_encoder_backward */
  int attn_o_resid;               /* as smbv_mim_cfg.attn_o_resid (train = 1 only) */
} smbv_encoder_cfg;

typedef struct smbv_encoder_tensors {
  float *pe_w, *pe_b;             /* embeddings.patch_embeddings.projection */
  smbv_layer_ptrs enc[SMBV_MAX_LAYERS];
  float *lnf_w, *lnf_b;           /* layernorm.{weight,bias} or NULL */
} smbv_encoder_tensors;

typedef struct smbv_encoder_io {
  const float* pixel_values;
  const int32_t* sel_idx;         /* int32 [B,Nsel] ascending token ids to process (0..N-1 each when there is no mask) */
  const float* pos_enc;           /* fp32 [N,d] sin-cos table */
  float* last_hidden_state;       /* fp32 [B*Nsel, d] */
} smbv_encoder_io;

size_t smbv_encoder_workspace_bytes(const smbv_encoder_cfg* cfg);
int smbv_encoder_forward(const smbv_encoder_cfg* cfg, const smbv_encoder_tensors* params, const smbv_encoder_io* io, void* ws, size_t ws_bytes,
                         smbv_stream stream);
/* output_hidden_states=True (spec :452-457, 478-483: the encoder collects the embedding output and every layer's output, NOT the final LayerNorm's): after a
 * forward with cfg.train = 1, out[l] (l = 0 .. layers; n_out = layers + 1) points at the fp32 [B*Nsel, d] residual stream the workspace holds in front of layer l
 * (out[layers] = the last layer's output).  Pointers into `ws`: valid until the workspace is rewritten by the next forward; no launch, no synchronisation.
 * The inference forward (cfg.train = 0) shares one activation set between its layers and is refused. */
int smbv_encoder_hidden_states(const smbv_encoder_cfg* cfg, void* ws, size_t ws_bytes, float** out, int n_out);
/* Fine-tuning (src/run_classification.py): gradients of every tensor in `grads` (overwritten) from d(last_hidden_state) fp32 [B*Nsel, d];
 * same cfg (train = 1), io and workspace as the forward that preceded it. */
int smbv_encoder_backward(const smbv_encoder_cfg* cfg, const smbv_encoder_tensors* params, const smbv_encoder_tensors* grads, const smbv_encoder_io* io,
                          const float* d_last_hidden_state, void* ws, size_t ws_bytes, smbv_stream stream);

/* ---- fused optimizer tail: clip_grad_norm_ + AdamW as HF Trainer runs them for the reference
 *      (site-packages/transformers/trainer.py:2539, :1785; --max_grad_norm / adamw_torch of scripts/training/run_mim.sh) ---- */

/* One contiguous piece (<= 2^31 elements; 16-byte aligned pointers) of one parameter tensor. The table lives in DEVICE memory. */
typedef struct smbv_adam_chunk {
  float* p; const float* g; float* m; float* v;   /* parameter, gradient, exp_avg, exp_avg_sq */
  int32_t n;                                     /* elements in this chunk */
  int32_t group;                                 /* index into smbv_adam_hyper.lr / .weight_decay (parameter group) */
} smbv_adam_chunk;

typedef struct smbv_adam_hyper {
  float lr[8], weight_decay[8];
  float beta1, beta2, eps;
  float bias_corr1, bias_corr2;                  /* 1 - beta^t, computed by the host */
  float max_grad_norm;                           /* <= 0: no clipping */
} smbv_adam_hyper;

/* out_sq[0] = sum over all chunks of g^2 (deterministic two-stage reduction); partial: fp32 [nchunks] scratch. */
int smbv_grad_sqnorm(const smbv_adam_chunk* chunks_dev, int nchunks, float* partial, float* out_sq, smbv_stream stream);
/* p, m, v updated in place; the clip coefficient min(1, max_norm/(sqrt(*gnorm_sq)+1e-6)) is applied on the fly (g untouched). */
int smbv_adamw_step(const smbv_adam_chunk* chunks_dev, int nchunks, const smbv_adam_hyper* hyper /* host */, const float* gnorm_sq,
                    smbv_stream stream);

/* ---- V-JEPA building blocks (src/run_vjepa.py, src/models/vjepa/modeling_vjepa.py) ------------------------------------------- */

/* 3-axis rotary embedding of the q (column 0) and k (column k_off) sections of the fused QKV matrix bf16 [rows, ld], in place
 * (modeling_vjepa.py:204-228 rotate_queries_or_keys with its tiled cos/sin, :277-326 position ids -> frame/height/width, three
 * segments of 2*((Dh/3)/2) dims per head).  head_stride: columns between heads (64 when heads are padded).  pos: int32 [rows] token
 * position ids or NULL (= row % S).  table: fp32 [P][2][seg/2] = cos | sin of position * 10000^(-j/(seg/2)).  transpose != 0 applies
 * the transposed map (gradient). */
int smbv_rope3d(smbv_bf16* qkv, int ld, int k_off, long rows, int S, int H, int head_stride, int Dh, int grid, const int32_t* pos,
                const float* table, int transpose, smbv_stream stream);
/* The same map as a by-product of the launches either side of it (ABI 3004; what the V-JEPA steps run since round 5 - the stand-alone launches were 2.2 ms of
 * kernel time per step, 36 passes over the q | k sections).  One description of the map for both: */
typedef struct smbv_rope3d_cfg {
  const float* table;              /* as smbv_rope3d */
  const int32_t* pos;              /* int32 [rows] or NULL (= row % S) */
  int S, grid, Dh, head_stride;
  int table_rows;                  /* P: the table covers frame / height / width ids 0 .. P-1 and every id of the call is below it (the fused forms stage the
                                      table into LDS and decode positions without integer division: P * seg * 4 <= 32 KiB and P * grid^4 < 2^32; otherwise, or with 0 =
                                      unknown, the map runs as the smbv_rope3d launch) */
} smbv_rope3d_cfg;
/* smbv_gemm_nt whose bf16 output [M, N] holds q in columns [0, wa) and k in [wa, 2 wa) (head slots of head_stride columns; N >= 2 wa, e.g. the fused QKV
 * product with N = 3 wa): the forward map is applied to those sections of the bf16-ROUNDED product in the epilogue - the stored matrix is bitwise what
 * smbv_gemm_nt followed by smbv_rope3d(out, ld, wa, ...) leaves.  epi: bias and out_bf16 only (SMBV_ERR_ARG otherwise); wa and head_stride multiples of 8.
 * Products the register-epilogue kernel does not take run as exactly those two launches. */
int smbv_gemm_nt_rope(const smbv_bf16* A, int lda, const smbv_bf16* B, int ldb, int M, int N, int K, const smbv_gemm_epi* epi, const smbv_rope3d_cfg* rope,
                      int wa, smbv_stream stream);
/* apply_masks (:542-556): dst[r,:] = src[batch(r)*src_batch_rows + idx[r], :]; scatter is the inverse placement; fill writes `vec`. */
int smbv_gather_rows_idx_f32(const float* src, float* dst, const int32_t* idx, long nidx, int per_batch, long src_batch_rows, int D, smbv_stream stream);
int smbv_scatter_rows_idx_f32(const float* src, float* dst, const int32_t* idx, long nidx, int per_batch, long dst_batch_rows, int D, smbv_stream stream);
int smbv_fill_rows_idx_f32(float* dst, const float* vec, const int32_t* idx, long nidx, int per_batch, long dst_batch_rows, int D, smbv_stream stream);
/* nn.L1Loss() (run_vjepa.py:108,136): loss = mean|pred - target|; dpred = sign(pred - target) * (*gscale) / n.  partial: fp32 [1024]. */
int smbv_l1_fwd(const float* pred, const float* target, size_t n, float* partial, float* loss, smbv_stream stream);
int smbv_l1_bwd(const float* pred, const float* target, size_t n, float* dpred, const float* gscale, smbv_stream stream);
/* MomentumEncoder.update (run_vjepa.py:94-98): chunk.p (target) = m * chunk.p + (1 - m) * chunk.g (online); one launch. */
int smbv_ema_update(const smbv_adam_chunk* chunks_dev, int nchunks, float momentum, smbv_stream stream);

/* ---- the V-JEPA online-model step: VJEPA2Model.forward with context/target masks (modeling_vjepa.py:1081-1163) and its autograd ---- */

typedef struct smbv_vjepa_cfg {
  int struct_bytes;                /* sizeof(smbv_vjepa_cfg) as the caller compiled it */
  int B, T, H, W;                  /* pixel_values_videos fp32 [B,T,1,H,W], H == W == crop_size */
  int ps, ts;                      /* patch_size, tubelet_size */
  int N, Nctx, Ntgt;               /* tokens per volume; context / target tokens per volume */
  int d, heads, ff, layers;        /* encoder: hidden_size, num_attention_heads, hidden_size*mlp_ratio, num_hidden_layers */
  int pd, pheads, pff, players;    /* predictor: pred_hidden_size, pred_num_attention_heads, ..., pred_num_hidden_layers */
  float ln_eps;                    /* config.layer_norm_eps (1e-6) */
  int n_masks;                     /* number of (context, target) mask pairs in the lists handed to VJEPA2Model.forward (0 or 1 = one pair).  apply_masks
                                      (modeling_vjepa.py:542-556) concatenates them along the batch: the predictor runs on P = n_masks * B sequences, sequence
                                      m * B + b pairing volume b with mask pair m; every per-sequence array below then has P rows instead of B */
  int enc_needed;                  /* 0 or N: the encoder's output is produced for every token (the reference's forward).  0 < enc_needed < N: the caller has
                                      ordered every volume's tokens (all_idx / enc_pos) so that the ones whose encoder OUTPUT it consumes come first - the
                                      context tokens for the online model, the target tokens for the frozen target encoder (run_vjepa.py:118-131 reads nothing
                                      else) - and the LAST encoder block computes queries, projection and MLP for those enc_needed rows only (keys / values: all
                                      N tokens; exact for the rows produced, as the last MIM decoder block).  enc_out and the rows ctx_idx names are then
                                      COMPACT: [B * enc_needed, d], row b * enc_needed + j.  Needs n_masks <= 1 */
  int pred_tgt_first;              /* 1: the caller placed the target tokens in slots 0 .. Ntgt-1 of every predictor sequence (tgt_slot < Ntgt, ctx_slot >= Ntgt;
                                      any order is the same function: pred_pos carries the positions, modeling_vjepa.py:718-746 sorts and unsorts).  The predictor's
                                      output is read at the target tokens only (:737-738), so its LAST block then computes those Ntgt rows only.  0: slots in any
                                      order (e.g. the reference's position-sorted one), every block over all Nctx + Ntgt rows */
  int attn_o_resid;                /* as smbv_mim_cfg.attn_o_resid.  The V-JEPA encoder at init scale is the case it exists for: attention over 20,480 near-identical
                                      tokens is uniform, dQ / dK are what survives a 1/sqrt(S) cancellation, and the 2^-9 rounding of a bf16 O inside delta does not
                                      cancel (10-15 % on some query / key weight gradients at 512x512x320 without it).  Set 0 for a forward that
                                      is never differentiated (the frozen target encoder) */
} smbv_vjepa_cfg;

/* fp32 tensors of one VJEPA2Layer: norm1, attention.{query,key,value,proj}.{weight,bias}, norm2, mlp.{fc1,fc2} */
typedef struct smbv_block_ptrs {
  float *ln1_w, *ln1_b, *q_w, *k_w, *v_w, *q_b, *k_b, *v_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} smbv_block_ptrs;

typedef struct smbv_vjepa_tensors {
  float *pe_w, *pe_b;                         /* encoder.embeddings.patch_embeddings.proj_3d */
  smbv_block_ptrs enc[SMBV_MAX_LAYERS];       /* encoder.layer.{i} */
  float *enc_ln_w, *enc_ln_b;                 /* encoder.layernorm */
  float *pemb_w, *pemb_b;                     /* predictor.embeddings.predictor_embeddings [pd,d] */
  float *mask_token;                          /* the selected row of predictor.embeddings.mask_tokens: [pd] */
  smbv_block_ptrs pred[SMBV_MAX_LAYERS];      /* predictor.layer.{i} */
  float *pred_ln_w, *pred_ln_b;               /* predictor.layernorm */
  float *pproj_w, *pproj_b;                   /* predictor.proj [d,pd] */
} smbv_vjepa_tensors;

typedef struct smbv_vjepa_io {
  const float* pixel_values;       /* fp32 [B,T,1,H,W] */
  const int32_t* all_idx;          /* int32 [B,N] brick (token) id of every encoder row: 0..N-1 per volume in the natural order, any permutation of it otherwise */
  const int32_t* ctx_idx;          /* int32 [P,Nctx] ROW of enc_out ([B*N] rows) each context token comes from: b * N + token id; distinct within a sequence */
  const int32_t* tgt_idx;          /* int32 [P,Ntgt] target token ids (not read by the library: the caller's target_hidden_state gather uses them) */
  const int32_t* pred_pos;         /* int32 [P,Nctx+Ntgt] position ids of the predictor sequence, sorted (modeling_vjepa.py:718-719) */
  const int32_t* ctx_slot;         /* int32 [P,Nctx] row of each context token in the sorted predictor sequence */
  const int32_t* tgt_slot;         /* int32 [P,Ntgt] row of each target token in the sorted predictor sequence */
  const float* rope_enc;           /* fp32 [P][2][seg/2] cos|sin table for the encoder head_dim (see smbv_rope3d); P >= max(T / ts, H / ps, W / ps): the steps
                                      stage that many rows into LDS for the rotary map they apply inside the QKV product and the attention backward */
  const float* rope_pred;          /* same for the predictor head_dim */
  float* enc_out;                  /* out: fp32 [B*N, d] last_hidden_state (after encoder.layernorm) */
  float* pred_out;                 /* out: fp32 [P*Ntgt, d] predictor_output.last_hidden_state (NULL: encoder only = skip_predictor) */
  const float* d_pred;             /* backward: fp32 [P*Ntgt, d] gradient w.r.t. pred_out */
  const float* d_enc;              /* backward, optional: fp32 [B*N, d] gradient w.r.t. enc_out (VJEPA2Model's last_hidden_state and the mask gathers taken from it,
                                      modeling_vjepa.py:1137-1163).  With d_pred == NULL the predictor is skipped (its gradients are left untouched: zero them).
                                      Not with enc_needed */
  const int32_t* enc_pos;          /* int32 [B,N] RoPE position id of every encoder row, or NULL = row index within the volume (the natural order).  With a
                                      permuted token order (enc_needed) this is all_idx: attention is permutation-equivariant once RoPE takes explicit ids */
} smbv_vjepa_io;

size_t smbv_vjepa_workspace_bytes(const smbv_vjepa_cfg* cfg);
int smbv_vjepa_forward(const smbv_vjepa_cfg* cfg, const smbv_vjepa_tensors* params, const smbv_vjepa_io* io, void* ws, size_t ws_bytes,
                       smbv_stream stream);
/* events / n_events: optional hipEvent_t array recorded as gradient groups complete (as smbv_mim_backward's: completion in array order), in this
 * order: [0] predictor.proj + predictor.layernorm, [1..players] predictor layers (last first), [players+1] mask tokens + predictor_embeddings,
 * [players+2] encoder.layernorm, [players+3 .. players+2+layers] encoder layers (last first), [last] patch embedding. */
int smbv_vjepa_backward(const smbv_vjepa_cfg* cfg, const smbv_vjepa_tensors* params, const smbv_vjepa_tensors* grads, const smbv_vjepa_io* io,
                        void* ws, size_t ws_bytes, void** events, int n_events, smbv_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* SMBV_H */
