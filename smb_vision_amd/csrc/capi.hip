// extern "C" surface of libsmbv.so (declared in include/smbv.h) + error plumbing.
#include "smbv_common.h"
#include "smbv_kernels.h"
#include "block.h"
#include "../../include/smbv.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace smbv {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return SMBV_OK;
}

static GemmEpi to_epi(const smbv_gemm_epi* s) {
  GemmEpi e;
  if (!s) return e;
  e.alpha = s->alpha;
  e.bias = s->bias;
  e.rowadd = s->rowadd; e.rowidx = s->rowidx; e.ldr = s->ldr;
  e.dgelu_pre = s->dgelu_pre; e.ld_pre = s->ld_pre;
  e.act_gelu = s->act_gelu; e.aux_pre = s->aux_pre; e.ld_aux = s->ld_aux;
  e.out_bf16 = s->out_bf16; e.ld_ob = s->ld_ob;
  e.out_f32 = s->out_f32; e.ld_of = s->ld_of;
  e.grp_in = s->grp_in; e.grp_out = s->grp_out;
  return e;
}

int g_variant = 0;  // smbv_debug_set_variant (include/smbv.h): kernel selection for A/B tools and equivalence tests; 0 on the product path
// second A/B word (the first one's 31 bits are taken): environment SMBV_DEBUG_AB, read once at load; 0 / unset on the product path.  Bits: 1 the inline
// block backward keeps round 3's join in front of its QKV products; 2 adamw_kernel walks its chunk table front to back; 4 the last grouped weight-gradient
// launch of a backward runs whole; 8 V-JEPA's rotary embedding as launches of its own (tools/step_ab2.sh)
int g_ab = [] { const char* e = getenv("SMBV_DEBUG_AB"); return e ? atoi(e) : 0; }();

}  // namespace smbv

using namespace smbv;

extern "C" {

int smbv_version(void) { return 3006; }  // 3000: O_lo in smbv_attention_fwd / _bwd, attn_o_resid in the *_cfg structs; see include/smbv.h
void smbv_debug_set_variant(int bits) { smbv::g_variant = bits; }
int smbv_device_status(int clear) { return smbv::device_status(clear != 0); }
int smbv_debug_pair_state(int op, smbv_stream stream) { return smbv::pair_state_debug(op, (hipStream_t)stream); }

int smbv_last_error(char* buf, size_t len) {
  const size_t n = strlen(g_err);
  if (buf && len) {
    const size_t c = n < len - 1 ? n : len - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return (int)n;
}

int smbv_gemm_nt(const smbv_bf16* A, int lda, const smbv_bf16* B, int ldb, int M, int N, int K, const smbv_gemm_epi* epi, smbv_stream stream) {
  if (!A || !B || !epi || (!epi->out_bf16 && !epi->out_f32)) { set_error("smbv_gemm_nt: null operand / no output"); return SMBV_ERR_ARG; }
  return gemm_nt(A, lda, B, ldb, M, N, K, to_epi(epi), (hipStream_t)stream);
}
int smbv_gemm_nt_splitk(const smbv_bf16* A, int lda, const smbv_bf16* B, int ldb, int M, int N, int K, const smbv_gemm_epi* epi, void* ws, size_t ws_bytes,
                        smbv_stream stream) {
  if (!A || !B || !epi || (!epi->out_bf16 && !epi->out_f32) || !ws) { set_error("smbv_gemm_nt_splitk: null operand / no output / no workspace"); return SMBV_ERR_ARG; }
  GemmEpi e = to_epi(epi);
  e.splitk_ws = ws;
  e.splitk_ws_bytes = ws_bytes;
  if (!gemm_nt_splitk_wanted(lda, ldb, M, N, K, e)) {
    set_error("smbv_gemm_nt_splitk: %d x %d x %d is not a product the split-K form takes (or the workspace is short / misaligned: %zu < %zu)", M, N, K, ws_bytes,
              gemm_nt_splitk_workspace_bytes(M, N));
    return SMBV_ERR_SHAPE;
  }
  hipError_t he = hipMemsetAsync(ws, 0, SPLITK_SYNC_BYTES, (hipStream_t)stream);
  if (he != hipSuccess) { set_error("smbv_gemm_nt_splitk: memset: %s", hipGetErrorString(he)); return (int)he; }
  return gemm_nt(A, lda, B, ldb, M, N, K, e, (hipStream_t)stream);
}
size_t smbv_gemm_nt_splitk_workspace_bytes(int M, int N) { return gemm_nt_splitk_workspace_bytes(M, N); }
int smbv_gemm_nt_rope(const smbv_bf16* A, int lda, const smbv_bf16* B, int ldb, int M, int N, int K, const smbv_gemm_epi* epi, const smbv_rope3d_cfg* rope, int wa,
                      smbv_stream stream) {
  if (!A || !B || !epi || !epi->out_bf16 || !rope || !rope->table) { set_error("smbv_gemm_nt_rope: null operand / no bf16 output / no table"); return SMBV_ERR_ARG; }
  GemmEpi e = to_epi(epi);
  e.rope_table = rope->table; e.rope_pos = rope->pos; e.rope_S = rope->S; e.rope_grid = rope->grid; e.rope_Dh = rope->Dh; e.rope_hs = rope->head_stride; e.rope_wa = wa; e.rope_rows = rope->table_rows;
  return gemm_nt(A, lda, B, ldb, M, N, K, e, (hipStream_t)stream);
}
int smbv_gemm_tn(const smbv_bf16* X, int ldx, const smbv_bf16* Y, int ldy, int M, int N1, int N2, float* out, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!X || !Y || !out) { set_error("smbv_gemm_tn: null operand"); return SMBV_ERR_ARG; }
  return gemm_tn(X, ldx, Y, ldy, M, N1, N2, out, ws, ws_bytes, (hipStream_t)stream);
}
size_t smbv_gemm_tn_workspace_bytes(int M, int N1, int N2) { return gemm_tn_workspace_bytes(M, N1, N2); }
int smbv_gemm_tn_bias(const smbv_bf16* X, int ldx, const smbv_bf16* Y, int ldy, int M, int N1, int N2, float* out, float* xsum, void* ws, size_t ws_bytes,
                      smbv_stream stream) {
  if (!X || !Y || !out || !xsum) { set_error("smbv_gemm_tn_bias: null operand"); return SMBV_ERR_ARG; }
  TnColsum cs;
  cs.dst[0] = xsum;
  cs.seg = N1;
  return gemm_tn(X, ldx, Y, ldy, M, N1, N2, out, ws, ws_bytes, (hipStream_t)stream, &cs);
}

static_assert(sizeof(smbv_tn_job) == sizeof(smbv::TnJob), "smbv_tn_job mirrors TnJob");
int smbv_gemm_tn_grouped(const smbv_tn_job* jobs, int njobs, smbv_stream stream) {
  if (!jobs || njobs <= 0 || njobs > TN_GROUP_MAX) { set_error("smbv_gemm_tn_grouped: 1..%d jobs", TN_GROUP_MAX); return SMBV_ERR_ARG; }
  TnJob j[TN_GROUP_MAX];
  memcpy(j, jobs, sizeof(TnJob) * njobs);
  return gemm_tn_grouped(j, njobs, (hipStream_t)stream);
}
static int tn_group_tiles_256(const smbv_tn_job* jobs, int njobs) {  // 0: not a launch the split form takes
  int tiles = 0;
  for (int i = 0; i < njobs; i++) {
    if ((jobs[i].N1 & 255) || (jobs[i].N2 & 255) || jobs[i].M < 256) return 0;
    tiles += (jobs[i].N1 / 256) * (jobs[i].N2 / 256);
  }
  return tiles <= 128 ? tiles : 0;
}
size_t smbv_gemm_tn_grouped_split_workspace_bytes(const smbv_tn_job* jobs, int njobs) {
  if (!jobs || njobs <= 0 || njobs > TN_GROUP_MAX) return 0;
  const int tiles = tn_group_tiles_256(jobs, njobs);
  return tiles ? gemm_tn_grouped_split_bytes(tiles) : 0;
}
int smbv_gemm_tn_grouped_split(const smbv_tn_job* jobs, int njobs, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!jobs || njobs <= 0 || njobs > TN_GROUP_MAX) { set_error("smbv_gemm_tn_grouped_split: 1..%d jobs", TN_GROUP_MAX); return SMBV_ERR_ARG; }
  const size_t need = smbv_gemm_tn_grouped_split_workspace_bytes(jobs, njobs);
  if (!need) { set_error("smbv_gemm_tn_grouped_split: every N1, N2 a multiple of 256, every M >= 256, at most 128 tiles"); return SMBV_ERR_SHAPE; }
  if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) { set_error("smbv_gemm_tn_grouped_split: workspace too small or not 256-byte aligned (%zu < %zu)", ws_bytes, need); return SMBV_ERR_WORKSPACE; }
  TnJob j[TN_GROUP_MAX];
  memcpy(j, jobs, sizeof(TnJob) * njobs);
  return gemm_tn_grouped(j, njobs, (hipStream_t)stream, ws, ws_bytes);
}

int smbv_layernorm_fwd(const float* x, const float* gamma, const float* beta, smbv_bf16* y, float* mean, float* rstd, int rows, int D, float eps,
                       long grp_in, long grp_out, long row_off, smbv_stream stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) { set_error("smbv_layernorm_fwd: null operand"); return SMBV_ERR_ARG; }
  return layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, eps, grp_in, grp_out, row_off, (hipStream_t)stream);
}
int smbv_layernorm_bwd(const smbv_bf16* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                       smbv_bf16* dx_bf16, float* dgamma, float* dbeta, int rows, int D, long grp_in, long grp_out, long row_off, void* ws,
                       size_t ws_bytes, smbv_stream stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dgamma || !dbeta) { set_error("smbv_layernorm_bwd: null operand"); return SMBV_ERR_ARG; }
  return layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, dx_bf16, dgamma, dbeta, rows, D, grp_in, grp_out, row_off, ws, ws_bytes, (hipStream_t)stream);
}
size_t smbv_layernorm_bwd_workspace_bytes(int rows, int D) { return layernorm_bwd_workspace_bytes(rows, D); }

int smbv_attention_fwd(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, smbv_bf16* O, smbv_bf16* O_lo, int ldo, float* LSE, int B, int S, int H,
                       int Dh, float scale, int Sq, int q_off, smbv_stream stream) {
  if (!Q || !K || !V || !O || !LSE) { set_error("smbv_attention_fwd: null operand"); return SMBV_ERR_ARG; }
  return attention_fwd(Q, K, V, ld, O, ldo, LSE, B, S, H, Dh, scale, Sq, q_off, (hipStream_t)stream, O_lo);
}
size_t smbv_attention_fp8_workspace_bytes(int B, int S, int H) { return attention_fp8_workspace_bytes(B, S, H); }
int smbv_attention_fwd_fp8(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, smbv_bf16* O, int ldo, float* LSE, int B, int S, int H, int Dh,
                           float scale, int Sq, int q_off, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!Q || !K || !V || !O || !LSE) { set_error("smbv_attention_fwd_fp8: null operand"); return SMBV_ERR_ARG; }
  return attention_fwd_fp8(Q, K, V, ld, O, ldo, LSE, B, S, H, Dh, scale, Sq, q_off, ws, ws_bytes, (hipStream_t)stream);
}
int smbv_attention_bwd(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, const smbv_bf16* O, const smbv_bf16* O_lo, const smbv_bf16* dO, int ldo,
                       const float* LSE, float* delta, smbv_bf16* dQ, smbv_bf16* dK, smbv_bf16* dV, int ldd, int B, int S, int H, int Dh, float scale,
                       int parts, int Sq, int q_off, smbv_stream stream) {
  if (!Q || !K || !V || !O || !dO || !LSE || !delta || !dQ || !dK || !dV) { set_error("smbv_attention_bwd: null operand"); return SMBV_ERR_ARG; }
  SMBV_CHECK_DEVICE_STATE();
  return attention_bwd(Q, K, V, ld, O, dO, ldo, LSE, delta, dQ, dK, dV, ldd, B, S, H, Dh, scale, parts, Sq, q_off, (hipStream_t)stream, O_lo);
}

size_t smbv_attention_bwd_split_workspace_bytes(int B, int S, int H, int Dh) { return (Dh == 64 || Dh == 32) ? attention_bwd_split_workspace_max(B, S, H, Dh) : 0; }
int smbv_attention_bwd_split(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, const smbv_bf16* O, const smbv_bf16* O_lo, const smbv_bf16* dO, int ldo,
                             const float* LSE, float* delta, smbv_bf16* dQ, smbv_bf16* dK, smbv_bf16* dV, int ldd, int B, int S, int H, int Dh, float scale,
                             int parts, int Sq, int q_off, void* ws, size_t ws_bytes, void* fp8_ws, size_t fp8_ws_bytes, smbv_stream stream) {
  if (!Q || !K || !V || !O || !dO || !LSE || !delta || !dQ || !dK || !dV) { set_error("smbv_attention_bwd_split: null operand"); return SMBV_ERR_ARG; }
  if (fp8_ws && (Dh != 64 || (parts & 6) != 6)) { set_error("smbv_attention_bwd_split: the fp8 query-owner blocks need head_dim 64 and both gradient parts"); return SMBV_ERR_SHAPE; }
  if (ws && (((uintptr_t)ws & 255) || ws_bytes < smbv_attention_bwd_split_workspace_bytes(B, S, H, Dh))) {
    set_error("smbv_attention_bwd_split: workspace short or misaligned (%zu < %zu)", ws_bytes, smbv_attention_bwd_split_workspace_bytes(B, S, H, Dh));
    return SMBV_ERR_WORKSPACE;
  }
  SMBV_CHECK_DEVICE_STATE();
  return attention_bwd(Q, K, V, ld, O, dO, ldo, LSE, delta, dQ, dK, dV, ldd, B, S, H, Dh, scale, parts, Sq, q_off, (hipStream_t)stream, O_lo, fp8_ws, fp8_ws_bytes, ws, ws_bytes);
}

int smbv_attention_bwd_rope(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, const smbv_bf16* O, const smbv_bf16* O_lo, const smbv_bf16* dO, int ldo,
                            const float* LSE, float* delta, smbv_bf16* dQ, smbv_bf16* dK, smbv_bf16* dV, int ldd, int B, int S, int H, int Dh, float scale,
                            int parts, int Sq, int q_off, void* ws, size_t ws_bytes, void* fp8_ws, size_t fp8_ws_bytes, const smbv_rope3d_cfg* rope, smbv_stream stream) {
  if (!Q || !K || !V || !O || !dO || !LSE || !delta || !dQ || !dK || !dV || !rope || !rope->table) { set_error("smbv_attention_bwd_rope: null operand"); return SMBV_ERR_ARG; }
  if ((parts & 6) != 6 || rope->head_stride != Dh || rope->Dh > Dh || (!rope->pos && rope->S != S) || dK <= dQ || dK - dQ >= ldd) {
    set_error("smbv_attention_bwd_rope: both gradient parts; head_stride = the attention's head slot; pos == NULL means positions 0..S-1; dQ | dK sections of one matrix");
    return SMBV_ERR_ARG;
  }
  if (fp8_ws && Dh != 64) { set_error("smbv_attention_bwd_rope: the fp8 query-owner blocks need head_dim 64"); return SMBV_ERR_SHAPE; }
  if (ws && (((uintptr_t)ws & 255) || ws_bytes < smbv_attention_bwd_split_workspace_bytes(B, S, H, Dh))) {
    set_error("smbv_attention_bwd_rope: workspace short or misaligned (%zu < %zu)", ws_bytes, smbv_attention_bwd_split_workspace_bytes(B, S, H, Dh));
    return SMBV_ERR_WORKSPACE;
  }
  SMBV_CHECK_DEVICE_STATE();
  RopeBwd rb;
  rb.table = rope->table; rb.pos = rope->pos; rb.grid = rope->grid; rb.Dh = rope->Dh; rb.rows = rope->table_rows;
  return attention_bwd(Q, K, V, ld, O, dO, ldo, LSE, delta, dQ, dK, dV, ldd, B, S, H, Dh, scale, parts, Sq, q_off, (hipStream_t)stream, O_lo, fp8_ws, fp8_ws_bytes, ws, ws_bytes, &rb);
}

size_t smbv_attention_bwd_fp8_workspace_bytes(int B, int S, int H) { return attention_bwd_fp8_workspace_bytes(B, S, H); }
int smbv_attention_bwd_fp8(const smbv_bf16* Q, const smbv_bf16* K, const smbv_bf16* V, int ld, const smbv_bf16* O, const smbv_bf16* O_lo, const smbv_bf16* dO, int ldo,
                           const float* LSE, float* delta, smbv_bf16* dQ, smbv_bf16* dK, smbv_bf16* dV, int ldd, int B, int S, int H, int Dh, float scale,
                           int parts, int Sq, int q_off, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!Q || !K || !V || !O || !dO || !LSE || !delta || !dQ || !dK || !dV || !ws) { set_error("smbv_attention_bwd_fp8: null operand"); return SMBV_ERR_ARG; }
  if (Dh != 64 || (parts & 6) != 6) { set_error("smbv_attention_bwd_fp8: head_dim 64 and both gradient parts (parts & 6 == 6) only"); return SMBV_ERR_SHAPE; }
  SMBV_CHECK_DEVICE_STATE();
  return attention_bwd(Q, K, V, ld, O, dO, ldo, LSE, delta, dQ, dK, dV, ldd, B, S, H, Dh, scale, parts, Sq, q_off, (hipStream_t)stream, O_lo, ws, ws_bytes);
}

int smbv_prepare_volume(const void* src, int src_dtype, float* dst, int B, int H0, int W0, int D0, int H, int W, int D, float a_min, float a_max,
                        float b_min, float b_max, int clip, smbv_stream stream) {
  if (!src || !dst) { set_error("smbv_prepare_volume: null operand"); return SMBV_ERR_ARG; }
  return prepare_volume(src, src_dtype, dst, B, H0, W0, D0, H, W, D, a_min, a_max, b_min, b_max, clip, (hipStream_t)stream);
}
int smbv_gather_bricks(const float* pix, const int32_t* idx, int rows, int per_batch, smbv_bf16* out, int T, int H, int W, int ts, int ps, smbv_stream stream) {
  if (!pix || !idx || !out) { set_error("smbv_gather_bricks: null operand"); return SMBV_ERR_ARG; }
  return gather_bricks(pix, idx, rows, per_batch, out, T, H, W, ts, ps, (hipStream_t)stream);
}
int smbv_brick_stats(const float* pix, const int32_t* idx, int rows, int per_batch, float* mean, float* rinv, int T, int H, int W, int ts, int ps, smbv_stream stream) {
  if (!pix || !idx || !mean || !rinv) { set_error("smbv_brick_stats: null operand"); return SMBV_ERR_ARG; }
  return brick_stats(pix, idx, rows, per_batch, mean, rinv, T, H, W, ts, ps, (hipStream_t)stream);
}
int smbv_mse_fwd(const float* pix, const int32_t* idx, int rows, int per_batch, const float* mean, const float* rinv, const smbv_bf16* logits,
                 float* row_partial, float* loss, int T, int H, int W, int ts, int ps, smbv_stream stream) {
  if (!pix || !idx || !mean || !rinv || !logits || !row_partial || !loss) { set_error("smbv_mse_fwd: null operand"); return SMBV_ERR_ARG; }
  return mse_fwd(pix, idx, rows, per_batch, mean, rinv, logits, row_partial, loss, T, H, W, ts, ps, (hipStream_t)stream);
}
int smbv_mse_bwd(const float* pix, const int32_t* idx, int rows, int per_batch, const float* mean, const float* rinv, const smbv_bf16* logits,
                 smbv_bf16* dlogits, const float* gscale, int T, int H, int W, int ts, int ps, smbv_stream stream) {
  if (!pix || !idx || !mean || !rinv || !logits || !dlogits) { set_error("smbv_mse_bwd: null operand"); return SMBV_ERR_ARG; }
  return mse_bwd(pix, idx, rows, per_batch, mean, rinv, logits, dlogits, gscale, T, H, W, ts, ps, (hipStream_t)stream);
}

int smbv_cast_f32_bf16(const float* src, smbv_bf16* dst, size_t n, smbv_stream stream) {
  if (!src || !dst) { set_error("smbv_cast_f32_bf16: null operand"); return SMBV_ERR_ARG; }
  return cast_f32_bf16(src, dst, n, (hipStream_t)stream);
}
int smbv_transpose_cast_f32_bf16(const float* src, smbv_bf16* dst, int R, int C, int ldd, smbv_stream stream) {
  if (!src || !dst) { set_error("smbv_transpose_cast_f32_bf16: null operand"); return SMBV_ERR_ARG; }
  return transpose_cast_f32_bf16(src, dst, R, C, ldd, (hipStream_t)stream);
}
int smbv_colsum_bf16(const smbv_bf16* X, int ld, int M, int N, float* out, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!X || !out) { set_error("smbv_colsum_bf16: null operand"); return SMBV_ERR_ARG; }
  return colsum_bf16(X, ld, M, N, out, 0, 0, 0, ws, ws_bytes, (hipStream_t)stream);
}
size_t smbv_colsum_workspace_bytes(int M, int N) { return colsum_workspace_bytes(M, N); }

}  // extern "C"
