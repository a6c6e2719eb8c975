// Internal C++ launch functions (one per kernel family).  Every function enqueues on `st`, never allocates device
// memory, never synchronises, and returns SMBV_OK or a negative SMBV_ERR_* / positive hipError_t.
#pragma once
#include "smbv_common.h"
#include <algorithm>
#include <cstddef>

namespace smbv {

// Epilogue of gemm_nt, applied to the fp32 accumulator tile before the store.
struct GemmEpi {
  float alpha = 1.0f;               // acc *= alpha
  const float* bias = nullptr;      // + bias[n]
  const float* rowadd = nullptr;    // + rowadd[(rowidx ? rowidx[m] : m) * ldr + n]   (residual stream / position table)
  const int* rowidx = nullptr;
  int ldr = 0;
  int radd_grp_in = 0, radd_grp_out = 0, radd_off = 0;  // rowadd row remap when rowidx == null: (m / in) * out + m % in + off
  const bf16_t* dgelu_pre = nullptr;  // *= gelu'(dgelu_pre[m * ld_pre + n])          (dgrad through GELU)
  int ld_pre = 0;
  int act_gelu = 0;                 // exact-erf GELU; when aux_pre != null the bf16 pre-activation is stored there
  bf16_t* aux_pre = nullptr;
  int ld_aux = 0;
  bf16_t* out_bf16 = nullptr;
  int ld_ob = 0;
  float* out_f32 = nullptr;
  int ld_of = 0;
  int grp_in = 0, grp_out = 0;      // output row remap: (m / grp_in) * grp_out + m % grp_in   (0 = identity)
  // row-dot by-product (the LDS-DMA kernel only, gemm_nt_rowdot_ok): dot_out[((m / dot_rows) * heads + n / dot_dh) * dot_rows + m % dot_rows] =
  // sum over the dot_dh columns of head n / dot_dh of bf16(out)[m, n] * dot_with[orow(m) * ld_dot + n]   (heads = N / dot_dh; dot_dh 64 or 32).
  // With out = dO (the attention output's gradient) and dot_with = O this is flash attention's delta = rowsum(dO * O), which the attention backward
  // otherwise computes in a launch of its own (attn_delta_kernel, 16 launches per MIM step)
  const bf16_t* dot_with = nullptr;
  const bf16_t* dot_with_lo = nullptr;  // optional second addend of the dot operand (same layout): dot over out * (dot_with + dot_with_lo)
  int ld_dot = 0;
  float* dot_out = nullptr;
  int dot_dh = 64, dot_rows = 0;
  int dot_heads = 0;                // (filled in by gemm_nt: N / dot_dh)
  // split-K scratch (gemm_nt_ring.hip, round 5): when set (gemm_nt_splitk_workspace_bytes, 256-byte aligned, its first 4 KiB ZERO at launch) gemm_nt runs a
  // long-K product whose 256 x 128 tiles do not fill the chip as two K halves per tile; null: never.  Only smbv_gemm_nt_splitk sets it: in the steps the
  // form measured SLOWER than the mixed-tile 128 x 128 launch (profiles/r05_splitk_ab.txt) and is not selected
  void* splitk_ws = nullptr;
  size_t splitk_ws_bytes = 0;
  // 3-axis rotary embedding of the q | k column sections on the way out (V-JEPA's fused QKV product, round 5): columns [0, 2 * rope_wa) in head slots of
  // rope_hs columns get rope3d's map - applied to the bf16-ROUNDED product, exactly what the rope3d launch behind the product computed before (bitwise the
  // same matrix) - in the ring kernel's epilogue (EPI 3); a product that kernel does not take runs plain and gemm_nt launches rope3d behind it.
  // Needs: out_bf16 only (no fp32 output, no activation), rope_wa and rope_hs multiples of 4
  const float* rope_table = nullptr;   // fp32 [P][2][seg / 2] cos | sin
  const int* rope_pos = nullptr;       // token position ids [M] (null: m % rope_S)
  int rope_S = 0, rope_grid = 0, rope_Dh = 0, rope_hs = 0, rope_wa = 0;
  int rope_rows = 0;                   // positions per axis the table covers (0: unknown - the map then runs as the launch behind the product)
  RopeGeo rope_geo;                    // (filled in by gemm_nt)
};
// split-K form of the ring kernel: is it the selection for this product, and what it needs
bool gemm_nt_splitk_wanted(int lda, int ldb, int M, int N, int K, const GemmEpi& e);
size_t gemm_nt_splitk_workspace_bytes(int M, int N);
static constexpr size_t SPLITK_SYNC_BYTES = 4096;   // flag words in front of the partial tiles
bool gemm_nt_rowdot_ok(int lda, int ldb, int M, int N, int K, const GemmEpi& e);

int gemm_nt(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e, hipStream_t st);
// second-generation NT kernel (gemm_nt_ring.hip): 256 x 128 tiles, LDS ring of K slices, register epilogue; gemm_nt dispatches to it
bool gemm_nt_ring_ok(int lda, int ldb, int M, int N, int K, const GemmEpi& e);
int gemm_nt_ring(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e, int slots, hipStream_t st, int dbg = 0);
int gemm_nt_ring_splitk(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e, hipStream_t st);
// optional by-product of gemm_tn: the column sums of X over the M rows (the bias gradient of the linear whose weight gradient this is),
// written as up to three segments of `seg` columns (fused q | k | v: three bias tensors; a null destination drops its segment)
struct TnColsum {
  float* dst[3] = {nullptr, nullptr, nullptr};
  int seg = 0;
};
int gemm_tn(const bf16_t* X, int ldx, const bf16_t* Y, int ldy, int M, int N1, int N2, float* out, void* ws, size_t ws_bytes, hipStream_t st,
            const TnColsum* colsum = nullptr);
size_t gemm_tn_workspace_bytes(int M, int N1, int N2);
// Grouped weight-gradient launch: up to TN_GROUP_MAX products out[N1,N2] = X[M,N1]^T Y[M,N2] as ONE grid, every output tile reduced over its product's
// whole token range (no slabs, no finishing launch); seg > 0: the column sums of X go straight to sum_dst[n / seg][n % seg] (null segment = dropped).
// All products of a launch use one tile size: 256 x 256 when every N1, N2 is a multiple of 256, else 128 x 128 (multiples of 128 required).
static constexpr int TN_GROUP_MAX = 8;
struct TnJob {
  const bf16_t* X; const bf16_t* Y; float* out;
  float* sum_dst[3];
  int ldx, ldy, M, N1, N2, seg;
  int tile0;  // (filled in by gemm_tn_grouped) first block of this job
};
struct TnGroup {
  TnJob job[TN_GROUP_MAX]; int njobs;
  int tiles;                    // > 0: split launch - 2 x tiles blocks, each tile's token range as two halves (gemm.hip)
  f32x4* scratch; unsigned* flags; unsigned* status;
  unsigned* ticket; unsigned ticket0;   // split launch: units are CLAIMED, in order, from a device counter (this launch's run starts at ticket0)
  unsigned want;                        // ... and its hand-off flags carry this launch's sequence number (never 0)
};
bool gemm_tn_grouped_ok(const TnJob* jobs, int njobs);
// split_ws (256-byte aligned, gemm_tn_grouped_split_bytes(tiles)): the launch runs every 256 x 256 tile's token range as two halves by two workgroups
int gemm_tn_grouped(const TnJob* jobs, int njobs, hipStream_t st, void* split_ws = nullptr, size_t split_ws_bytes = 0);
size_t gemm_tn_grouped_split_bytes(int tiles);

int layernorm_fwd(const float* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd, int rows, int D, float eps,
                  long grp_in, long grp_out, long row_off, hipStream_t st, float* y32 = nullptr);
int layernorm_bwd(const bf16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                  bf16_t* dx_bf16, float* dgamma, float* dbeta, int rows, int D, long grp_in, long grp_out, long row_off, void* ws,
                  size_t ws_bytes, hipStream_t st, bool launch_reduce = true);
int layernorm_bwd_reduce(const void* ws, int rows, int D, float* dgamma, float* dbeta, hipStream_t st);
size_t layernorm_bwd_workspace_bytes(int rows, int D);

// O_lo (optional, same layout as O): the forward also stores bf16(O_fp32 - O), the backward forms delta = rowsum(dO * (O + O_lo)) - the rounding of
// the stored bf16 output otherwise enters delta at 2^-9 relative, which a near-uniform softmax over many keys does not average out
int attention_fwd(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, bf16_t* O, int ldo, float* LSE, int B, int S, int H, int Dh,
                  float scale, int Sq, int qoff, hipStream_t st, bf16_t* O_lo = nullptr);
// the rotary map the q / k of this attention went through (attention_bwd: dQ / dK leave through its transpose; Dh = the REAL head_dim, <= the slot width)
struct RopeBwd {
  const float* table = nullptr;
  const int* pos = nullptr;   // position ids by row of the [B * S] token matrix; null: the token's index in its sequence
  int grid = 0, Dh = 0;
  int rows = 0;               // positions per axis the table covers (0: unknown - the map then runs as a launch behind the kernels)
};
int attention_bwd(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, const bf16_t* O, const bf16_t* dO, int ldo, const float* LSE,
                  float* delta, bf16_t* dQ, bf16_t* dK, bf16_t* dV, int ldd, int B, int S, int H, int Dh, float scale, int parts, int Sq, int qoff, hipStream_t st,
                  const bf16_t* O_lo = nullptr, void* f8ws = nullptr, size_t f8ws_bytes = 0, void* split_ws = nullptr, size_t split_ws_bytes = 0,
                  const RopeBwd* rope = nullptr);
// split_ws (256-byte aligned, attention_bwd_split_workspace_max bytes): the pair launch runs the last tiles of each role as two half sweeps (attention_bwd.hip SweepPart)
size_t attention_bwd_split_workspace_bytes(int Dh, int xk, int xq);
size_t attention_bwd_split_workspace_max(int B, int S, int H, int Dh);
// f8ws != null (head_dim 64, the one-launch pair): the query-owner blocks recompute S and dP on the block-scaled fp8 MFMA (BASELINE config 5)

// fp8 operands of the attention backward's query-owner blocks (attention_fp8.hip attention_bwd_fp8_prepare -> attention_bwd.hip attn_dq_fp8_block)
struct Fp8BwdOps {
  const unsigned char* base = nullptr;   // workspace: Q8 | K8 | V8 | dO8 (e4m3 rows [B*S, w]) | their E8M0 scales (Q, dO: [rows, w / 32]; K, V: [B, H][S][2])
  unsigned bytes = 0, offK = 0, offV = 0, offD = 0, offSQ = 0, offSK = 0, offSV = 0, offSD = 0;
  int w = 0;
};
size_t attention_bwd_fp8_workspace_bytes(int B, int S, int H);
int attention_bwd_fp8_prepare(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, const bf16_t* dO, int ldo, int B, int S, int H, float scale, void* ws,
                              size_t ws_bytes, hipStream_t st, Fp8BwdOps* ops);
// fp8 (OCP e4m3, block-scaled MFMA) forward: same contract as attention_fwd plus a caller-owned workspace for the quantised operands
size_t attention_fp8_workspace_bytes(int B, int S, int H);
int attention_fwd_fp8(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, bf16_t* O, int ldo, float* LSE, int B, int S, int H, int Dh, float scale,
                      int Sq, int qoff, void* ws, size_t ws_bytes, hipStream_t st);

int prepare_volume(const void* src, int src_dtype, float* dst, int B, int H0, int W0, int D0, int H, int W, int D, float a_min, float a_max,
                   float b_min, float b_max, int clip, hipStream_t st);
int gather_bricks(const float* pix, const int* idx, int rows, int per_batch, bf16_t* out, int T, int H, int W, int ts, int ps, hipStream_t st);
int brick_stats(const float* pix, const int* idx, int rows, int per_batch, float* mean, float* rinv, int T, int H, int W, int ts, int ps,
                hipStream_t st);
int mse_fwd(const float* pix, const int* idx, int rows, int per_batch, const float* mean, const float* rinv, const bf16_t* logits,
            float* row_partial, float* loss, int T, int H, int W, int ts, int ps, hipStream_t st);
int mse_bwd(const float* pix, const int* idx, int rows, int per_batch, const float* mean, const float* rinv, const bf16_t* logits,
            bf16_t* dlogits, const float* gscale, int T, int H, int W, int ts, int ps, hipStream_t st);

struct PrepJob {
  const float* src;
  void* dst;
  void* dst2; // kind 4: the transposed bf16 copy [C, ldd2] (null: none)
  int R, C, ldd;
  int kind;   // 0 cast, 1 cast + transpose, 2 fp32 copy of C elements, 3 zero-fill of C floats, 4 ONE read of the fp32 master -> bf16 copy (dst, may be null) AND
              // transposed bf16 copy (dst2, may be null), 64 x 64 tiles, 16-byte loads / 8-byte stores (round 5)
  int tile0;  // first block of this job
  int lds;    // source leading dimension (elements)
  int ldd2;   // leading dimension of dst2
};
static inline int prep_job_tiles(const PrepJob& j) {
  return j.kind == 4 ? (int)(((j.R + 63) / 64) * ((j.C + 63) / 64)) : (int)(((j.R + 31) / 32) * ((j.C + 31) / 32));
}
int prep_weights(const PrepJob* jobs_host, int njobs, hipStream_t st);  // jobs travel in the kernel arguments, 64 per launch
int cast_f32_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st);
int transpose_cast_f32_bf16(const float* src, bf16_t* dst, int R, int C, int ldd, hipStream_t st);
int fill_mask_rows(float* xfull, const float* mask_token, const float* pos, const int* mask_idx, int B, int nmask, int nvis, int D, hipStream_t st);
int gather_rows_bf16(const bf16_t* in, bf16_t* out, int rows, int D, long grp_in, long grp_out, long row_off, hipStream_t st);
int colsum_bf16(const bf16_t* X, int ld, int M, int N, float* out, long grp_in, long grp_out, long row_off, void* ws, size_t ws_bytes, hipStream_t st);
int colsum_f32(const float* X, int ld, int M, int N, float* out, long grp_in, long grp_out, long row_off, void* ws, size_t ws_bytes, hipStream_t st);
size_t colsum_workspace_bytes(int M, int N);
int reduce_cols(const float* part, int nrows, int ncols, float* out0, float* out1, int split, hipStream_t st);

int rope3d(bf16_t* qkv, int ld, int k_off, long rows, int S, int H, int hs, int Dh, int grid, const int* pos, const float* table, int transpose,
           hipStream_t st);
int gather_rows_idx_bf16(const bf16_t* src, bf16_t* dst, const int* idx, long nidx, int per_batch, long src_batch_rows, int D, hipStream_t st);
int scatter_rows_idx_bf16(const bf16_t* src, bf16_t* dst, const int* idx, long nidx, int per_batch, long dst_batch_rows, int D, hipStream_t st);
int add_f32_into_bf16(bf16_t* dst, const float* src, size_t n, hipStream_t st);  // dst += src
int scatter_add_rows_idx_bf16(const bf16_t* src, bf16_t* dst, const int* idx, long nidx, int D, hipStream_t st);  // dst[idx[r]] += src[r], idx = absolute rows
int gather_rows_idx_f32(const float* src, float* dst, const int* idx, long nidx, int per_batch, long src_batch_rows, int D, hipStream_t st);
int scatter_rows_idx_f32(const float* src, float* dst, const int* idx, long nidx, int per_batch, long dst_batch_rows, int D, hipStream_t st);
int fill_rows_idx_f32(float* dst, const float* vec, const int* idx, long nidx, int per_batch, long dst_batch_rows, int D, hipStream_t st);
int map_copy_f32(float* dst, const float* src, size_t n, long grp_in, long grp_out, hipStream_t st);

}  // namespace smbv
