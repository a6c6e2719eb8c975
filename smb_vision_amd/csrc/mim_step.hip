// Host-side sequencing of the whole MIM train step (VideoMAEForPreTraining.forward, spec
// src/models/videomae/modeling_videomae.py:753-908, and its autograd) over the kernels of this library.
// One C call enqueues the entire forward (or backward) on the caller's stream; nothing here synchronises or allocates.
//
// dtype policy (mirrors torch.autocast(bf16) as HF Trainer applies it): residual stream, LayerNorm
// statistics, softmax, MSE, all parameter gradients in fp32; GEMM / attention operands in bf16 with fp32 accumulation.
#include "smbv_common.h"
#include "smbv_kernels.h"
#include "../../include/smbv.h"
#include <vector>

namespace smbv {

struct LayerWs {
  // bf16 weight copies
  bf16_t *wqkv, *wqkv_t, *proj, *proj_t, *fc1, *fc1_t, *fc2, *fc2_t;
  float* bqkv;
  // saved activations
  bf16_t *ln1, *qkv, *attn_o, *ln2, *pre, *act;
  float *mean1, *rstd1, *mean2, *rstd2, *lse, *x_mid;
};

struct MimWs {
  bf16_t *pe_w, *e2d, *e2d_t, *head, *head_t;
  LayerWs enc[SMBV_MAX_LAYERS], dec[SMBV_MAX_LAYERS];
  bf16_t* a_patch;
  float* x[SMBV_MAX_LAYERS + 1];   // encoder residual stream [R, d]
  float* xd[SMBV_MAX_LAYERS + 1];  // decoder residual stream [Rd, dd]
  bf16_t* enc_out;
  bf16_t* lnf;
  float *meanf, *rstdf, *bmean, *brinv, *row_partial;
  // backward scratch
  float *g0, *g1, *delta;
  bf16_t *gb0, *gb1, *dlogits, *d_act, *d_ln, *d_o, *d_qkv, *d_vis, *gath0, *gath1;
  PrepJob* prep_table;  // device copy of the weight-preparation job list (uploaded by every forward)
  void* scratch;  // gemm_tn slabs / ln-bwd partials / colsum partials (max of all uses)
  size_t scratch_bytes;
  size_t total;
};

struct Carver {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

static void carve_layer(Carver& c, LayerWs& L, size_t rows, int w, int ff, int B, int H, int S) {
  L.wqkv = c.take<bf16_t>((size_t)3 * w * w);
  L.wqkv_t = c.take<bf16_t>((size_t)3 * w * w);
  L.proj = c.take<bf16_t>((size_t)w * w);
  L.proj_t = c.take<bf16_t>((size_t)w * w);
  L.fc1 = c.take<bf16_t>((size_t)ff * w);
  L.fc1_t = c.take<bf16_t>((size_t)ff * w);
  L.fc2 = c.take<bf16_t>((size_t)ff * w);
  L.fc2_t = c.take<bf16_t>((size_t)ff * w);
  L.bqkv = c.take<float>((size_t)3 * w);
  L.ln1 = c.take<bf16_t>(rows * w);
  L.qkv = c.take<bf16_t>(rows * 3 * w);
  L.attn_o = c.take<bf16_t>(rows * w);
  L.ln2 = c.take<bf16_t>(rows * w);
  L.pre = c.take<bf16_t>(rows * ff);
  L.act = c.take<bf16_t>(rows * ff);
  L.mean1 = c.take<float>(rows);
  L.rstd1 = c.take<float>(rows);
  L.mean2 = c.take<float>(rows);
  L.rstd2 = c.take<float>(rows);
  L.lse = c.take<float>((size_t)B * H * S);
  L.x_mid = c.take<float>(rows * w);
}

static int check_cfg(const smbv_mim_cfg& c) {
  if (c.B <= 0 || c.layers <= 0 || c.dlayers <= 0 || c.layers > SMBV_MAX_LAYERS || c.dlayers > SMBV_MAX_LAYERS) { set_error("mim: bad batch/layer counts"); return SMBV_ERR_SHAPE; }
  if (c.ps <= 0 || c.ts <= 0 || c.T % c.ts || c.H % c.ps || c.W % c.ps || (c.ps & 3)) { set_error("mim: volume %dx%dx%d not divisible by tubelet %d / patch %d", c.T, c.H, c.W, c.ts, c.ps); return SMBV_ERR_SHAPE; }
  const int N = (c.T / c.ts) * (c.H / c.ps) * (c.W / c.ps);
  if (c.N != N || c.Nvis <= 0 || c.Nmask <= 0 || c.Nvis + c.Nmask != N) { set_error("mim: token counts N=%d Nvis=%d Nmask=%d inconsistent with the volume (%d)", c.N, c.Nvis, c.Nmask, N); return SMBV_ERR_SHAPE; }
  if (c.d % c.heads || c.d / c.heads != 64 || c.dd % c.dheads || c.dd / c.dheads != 64) { set_error("mim: head_dim must be 64 (d=%d/%d, dd=%d/%d)", c.d, c.heads, c.dd, c.dheads); return SMBV_ERR_SHAPE; }
  if ((c.d & 7) || (c.dd & 7) || (c.ff & 7) || (c.dff & 7) || ((c.ts * c.ps * c.ps) & 7)) { set_error("mim: widths must be multiples of 8"); return SMBV_ERR_SHAPE; }
  return SMBV_OK;
}

static void carve(const smbv_mim_cfg& c, void* base, MimWs& w) {
  Carver cv{(char*)base};
  const int K = c.ts * c.ps * c.ps;
  const size_t R = (size_t)c.B * c.Nvis, Rd = (size_t)c.B * c.N, Rm = (size_t)c.B * c.Nmask;
  w.prep_table = cv.take<PrepJob>(16 + 16 * (size_t)(c.layers + c.dlayers));
  w.pe_w = cv.take<bf16_t>((size_t)c.d * K);
  w.e2d = cv.take<bf16_t>((size_t)c.dd * c.d);
  w.e2d_t = cv.take<bf16_t>((size_t)c.dd * c.d);
  w.head = cv.take<bf16_t>((size_t)K * c.dd);
  w.head_t = cv.take<bf16_t>((size_t)K * c.dd);
  for (int l = 0; l < c.layers; l++) carve_layer(cv, w.enc[l], R, c.d, c.ff, c.B, c.heads, c.Nvis);
  for (int l = 0; l < c.dlayers; l++) carve_layer(cv, w.dec[l], Rd, c.dd, c.dff, c.B, c.dheads, c.N);
  w.a_patch = cv.take<bf16_t>(R * K);
  for (int l = 0; l <= c.layers; l++) w.x[l] = cv.take<float>(R * c.d);
  for (int l = 0; l <= c.dlayers; l++) w.xd[l] = cv.take<float>(Rd * c.dd);
  w.enc_out = cv.take<bf16_t>(R * c.d);
  w.lnf = cv.take<bf16_t>(Rm * c.dd);
  w.meanf = cv.take<float>(Rm);
  w.rstdf = cv.take<float>(Rm);
  w.bmean = cv.take<float>(Rm);
  w.brinv = cv.take<float>(Rm);
  w.row_partial = cv.take<float>(Rm);
  // backward scratch, sized for the wider of encoder / decoder
  const size_t rows_max = std::max(R, Rd);
  const size_t gw = std::max(R * c.d, Rd * c.dd);
  const size_t gf = std::max(R * c.ff, Rd * c.dff);
  w.g0 = cv.take<float>(gw);
  w.g1 = cv.take<float>(gw);
  w.gb0 = cv.take<bf16_t>(gw);
  w.gb1 = cv.take<bf16_t>(gw);
  w.delta = cv.take<float>(std::max((size_t)c.B * c.heads * c.Nvis, (size_t)c.B * c.dheads * c.N));
  w.dlogits = cv.take<bf16_t>(Rm * K);
  w.d_act = cv.take<bf16_t>(gf);
  w.d_ln = cv.take<bf16_t>(gw);
  w.d_o = cv.take<bf16_t>(gw);
  w.d_qkv = cv.take<bf16_t>(3 * gw);
  w.d_vis = cv.take<bf16_t>(R * c.dd);
  w.gath0 = cv.take<bf16_t>(Rm * c.dd);
  w.gath1 = cv.take<bf16_t>(Rm * c.dd);
  size_t sb = 0;
  auto tn = [&](size_t M, int n1, int n2) { sb = std::max(sb, gemm_tn_workspace_bytes((int)M, n1, n2)); };
  tn(R, c.d, c.ff); tn(R, c.ff, c.d); tn(R, c.d, c.d); tn(R, 3 * c.d, c.d); tn(R, c.d, K); tn(R, c.dd, c.d);
  tn(Rd, c.dd, c.dff); tn(Rd, c.dff, c.dd); tn(Rd, c.dd, c.dd); tn(Rd, 3 * c.dd, c.dd); tn(Rm, K, c.dd);
  sb = std::max(sb, layernorm_bwd_workspace_bytes((int)rows_max, std::max(c.d, c.dd)));
  sb = std::max(sb, colsum_workspace_bytes((int)rows_max, std::max(std::max(c.ff, c.dff), std::max(3 * c.d, K))));
  w.scratch = cv.take<char>(sb);
  w.scratch_bytes = sb;
  w.total = (cv.off + 255) & ~(size_t)255;
}

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
#define HIPRC(x) do { hipError_t _e = (x); if (_e != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(_e)); return (int)_e; } } while (0)

struct PrepList {
  std::vector<PrepJob> jobs;
  int tiles = 0;
  void add(const float* src, void* dst, int R, int C, int ldd, int kind) {
    PrepJob j;
    j.src = src; j.dst = dst; j.R = R; j.C = C; j.ldd = ldd; j.kind = kind; j.tile0 = tiles; j.pad = 0;
    jobs.push_back(j);
    tiles += cdiv(R, 32) * cdiv(C, 32);
  }
  void cast(const float* src, bf16_t* dst, int R, int C) { add(src, dst, R, C, C, 0); }
  void tcast(const float* src, bf16_t* dst, int R, int C, int ldd) { add(src, dst, R, C, ldd, 1); }
};

static void prep_layer(PrepList& pl, const smbv_layer_ptrs& p, LayerWs& L, int w, int ff) {
  const size_t ww = (size_t)w * w;
  pl.cast(p.q_w, L.wqkv, w, w);
  pl.cast(p.k_w, L.wqkv + ww, w, w);
  pl.cast(p.v_w, L.wqkv + 2 * ww, w, w);
  pl.tcast(p.q_w, L.wqkv_t, w, w, 3 * w);
  pl.tcast(p.k_w, L.wqkv_t + w, w, w, 3 * w);
  pl.tcast(p.v_w, L.wqkv_t + 2 * w, w, w, 3 * w);
  pl.cast(p.proj_w, L.proj, w, w);
  pl.tcast(p.proj_w, L.proj_t, w, w, w);
  pl.cast(p.fc1_w, L.fc1, ff, w);
  pl.tcast(p.fc1_w, L.fc1_t, ff, w, ff);
  pl.cast(p.fc2_w, L.fc2, w, ff);
  pl.tcast(p.fc2_w, L.fc2_t, w, ff, w);
  pl.add(p.q_b, L.bqkv, 1, w, w, 2);
  pl.add(nullptr, L.bqkv + w, 1, w, w, 3);  // key has no bias (spec :261)
  pl.add(p.v_b, L.bqkv + 2 * w, 1, w, w, 2);
}

// one pre-LN block forward (spec :405-431) over `rows` = B*S tokens of width w
// Sq > 0: only the tokens qoff..qoff+Sq-1 of every sequence are needed downstream (last decoder block: spec :717-718 keeps the
// last N_mask tokens).  LN1/QKV still run on all tokens (every token is a key/value); attention queries, the output projection
// and the MLP run on the Sq tokens only.  x_mid / x_out keep the full [B*S, w] layout (rows outside the range are not written);
// ln2 / pre / act are compact [B*Sq, .].
static int layer_fwd(const smbv_layer_ptrs& p, LayerWs& L, const float* x_in, float* x_out, bf16_t* x_out_bf16, size_t rows, int w, int ff,
                     int B, int S, int H, float eps, MimWs& W, int Sq, int qoff, hipStream_t st) {
  if (Sq > 0) {
    const int Mq = B * Sq;
    RC(layernorm_fwd(x_in, p.ln1_w, p.ln1_b, L.ln1, L.mean1, L.rstd1, (int)rows, w, eps, 0, 0, 0, st));
    {
      GemmEpi e; e.bias = L.bqkv; e.out_bf16 = L.qkv; e.ld_ob = 3 * w;
      RC(gemm_nt(L.ln1, w, L.wqkv, w, (int)rows, 3 * w, w, e, st));
    }
    RC(attention_fwd(L.qkv, L.qkv + w, L.qkv + 2 * w, 3 * w, L.attn_o, w, L.lse, B, S, H, 64, 0.125f, Sq, qoff, st));
    const bf16_t* a_rows = L.attn_o + (size_t)qoff * w;  // contiguous for B == 1
    if (B > 1) { RC(gather_rows_bf16(L.attn_o, W.gath0, Mq, w, Sq, S, qoff, st)); a_rows = W.gath0; }
    {
      GemmEpi e; e.bias = p.proj_b; e.rowadd = x_in; e.ldr = w; e.radd_grp_in = Sq; e.radd_grp_out = S; e.radd_off = qoff;
      e.out_f32 = L.x_mid + (size_t)qoff * w; e.ld_of = w; e.grp_in = Sq; e.grp_out = S;
      RC(gemm_nt(a_rows, w, L.proj, w, Mq, w, w, e, st));
    }
    RC(layernorm_fwd(L.x_mid, p.ln2_w, p.ln2_b, L.ln2, L.mean2, L.rstd2, Mq, w, eps, Sq, S, qoff, st));
    {
      GemmEpi e; e.bias = p.fc1_b; e.act_gelu = 1; e.aux_pre = L.pre; e.ld_aux = ff; e.out_bf16 = L.act; e.ld_ob = ff;
      RC(gemm_nt(L.ln2, w, L.fc1, w, Mq, ff, w, e, st));
    }
    {
      GemmEpi e; e.bias = p.fc2_b; e.rowadd = L.x_mid; e.ldr = w; e.radd_grp_in = Sq; e.radd_grp_out = S; e.radd_off = qoff;
      e.out_f32 = x_out + (size_t)qoff * w; e.ld_of = w; e.grp_in = Sq; e.grp_out = S;
      RC(gemm_nt(L.act, ff, L.fc2, ff, Mq, w, ff, e, st));
    }
    return SMBV_OK;
  }
  RC(layernorm_fwd(x_in, p.ln1_w, p.ln1_b, L.ln1, L.mean1, L.rstd1, (int)rows, w, eps, 0, 0, 0, st));
  {
    GemmEpi e; e.bias = L.bqkv; e.out_bf16 = L.qkv; e.ld_ob = 3 * w;
    RC(gemm_nt(L.ln1, w, L.wqkv, w, (int)rows, 3 * w, w, e, st));
  }
  RC(attention_fwd(L.qkv, L.qkv + w, L.qkv + 2 * w, 3 * w, L.attn_o, w, L.lse, B, S, H, 64, 0.125f, 0, 0, st));
  {
    GemmEpi e; e.bias = p.proj_b; e.rowadd = x_in; e.ldr = w; e.out_f32 = L.x_mid; e.ld_of = w;
    RC(gemm_nt(L.attn_o, w, L.proj, w, (int)rows, w, w, e, st));
  }
  RC(layernorm_fwd(L.x_mid, p.ln2_w, p.ln2_b, L.ln2, L.mean2, L.rstd2, (int)rows, w, eps, 0, 0, 0, st));
  {
    GemmEpi e; e.bias = p.fc1_b; e.act_gelu = 1; e.aux_pre = L.pre; e.ld_aux = ff; e.out_bf16 = L.act; e.ld_ob = ff;
    RC(gemm_nt(L.ln2, w, L.fc1, w, (int)rows, ff, w, e, st));
  }
  {
    GemmEpi e; e.bias = p.fc2_b; e.rowadd = L.x_mid; e.ldr = w; e.out_f32 = x_out; e.ld_of = w; e.out_bf16 = x_out_bf16; e.ld_ob = w;
    RC(gemm_nt(L.act, ff, L.fc2, ff, (int)rows, w, ff, e, st));
  }
  return SMBV_OK;
}

// backward of one block. in: g (fp32) / gb (bf16) = grad wrt the block output; out: same pair overwritten with the grad
// wrt the block input. g2/gb2 are the ping-pong partners.
// backward of a block whose forward ran with a query range (see layer_fwd): g/gb hold the gradient of the Sq tokens in the full
// layout (zeros elsewhere); ln2/pre/act are compact.
static int layer_bwd_ranged(const smbv_layer_ptrs& p, const smbv_layer_ptrs& gr, LayerWs& L, MimWs& W, const float* x_in, float* g, bf16_t* gb,
                            float* g2, bf16_t* gb2, size_t rows, int w, int ff, int B, int S, int H, int Sq, int qoff, hipStream_t st) {
  const int M = (int)rows, Mq = B * Sq;
  const bf16_t* gbc = gb + (size_t)qoff * w;  // compact view of the ranged rows (contiguous for B == 1)
  if (B > 1) { RC(gather_rows_bf16(gb, W.gath0, Mq, w, Sq, S, qoff, st)); gbc = W.gath0; }
  // FC2
  RC(colsum_bf16(gbc, w, Mq, w, gr.fc2_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(gemm_tn(gbc, w, L.act, ff, Mq, w, ff, gr.fc2_w, W.scratch, W.scratch_bytes, st));
  {
    GemmEpi e; e.dgelu_pre = L.pre; e.ld_pre = ff; e.out_bf16 = W.d_act; e.ld_ob = ff;
    RC(gemm_nt(gbc, w, L.fc2_t, w, Mq, ff, w, e, st));
  }
  // FC1
  RC(colsum_bf16(W.d_act, ff, Mq, ff, gr.fc1_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(gemm_tn(W.d_act, ff, L.ln2, w, Mq, ff, w, gr.fc1_w, W.scratch, W.scratch_bytes, st));
  {
    GemmEpi e; e.out_bf16 = W.d_ln; e.ld_ob = w;
    RC(gemm_nt(W.d_act, ff, L.fc1_t, ff, Mq, w, ff, e, st));
  }
  // LN2 (+ residual) on the ranged rows; every other row of g2 is zero
  HIPRC(hipMemsetAsync(g2, 0, rows * w * sizeof(float), st));
  HIPRC(hipMemsetAsync(gb2, 0, rows * w * sizeof(bf16_t), st));
  RC(layernorm_bwd(W.d_ln, L.x_mid, L.mean2, L.rstd2, p.ln2_w, g, g2, gb2, gr.ln2_w, gr.ln2_b, Mq, w, Sq, S, qoff, W.scratch, W.scratch_bytes, st));
  // attention output projection
  const bf16_t* g2c = gb2 + (size_t)qoff * w;
  const bf16_t* aoc = L.attn_o + (size_t)qoff * w;
  if (B > 1) {
    RC(gather_rows_bf16(gb2, W.gath0, Mq, w, Sq, S, qoff, st));
    RC(gather_rows_bf16(L.attn_o, W.gath1, Mq, w, Sq, S, qoff, st));
    g2c = W.gath0; aoc = W.gath1;
  }
  RC(colsum_bf16(g2c, w, Mq, w, gr.proj_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(gemm_tn(g2c, w, aoc, w, Mq, w, w, gr.proj_w, W.scratch, W.scratch_bytes, st));
  {
    GemmEpi e; e.out_bf16 = W.d_o + (size_t)qoff * w; e.ld_ob = w; e.grp_in = Sq; e.grp_out = S;
    RC(gemm_nt(g2c, w, L.proj_t, w, Mq, w, w, e, st));
  }
  // attention: dQ only exists for the ranged queries (zero elsewhere); dK/dV for every token
  HIPRC(hipMemsetAsync(W.d_qkv, 0, rows * 3 * w * sizeof(bf16_t), st));
  RC(attention_bwd(L.qkv, L.qkv + w, L.qkv + 2 * w, 3 * w, L.attn_o, W.d_o, w, L.lse, W.delta, W.d_qkv, W.d_qkv + w, W.d_qkv + 2 * w, 3 * w, B, S,
                   H, 64, 0.125f, 7, Sq, qoff, st));
  // QKV
  RC(colsum_bf16(W.d_qkv, 3 * w, M, w, gr.q_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(colsum_bf16(W.d_qkv + 2 * w, 3 * w, M, w, gr.v_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  const size_t ww = (size_t)w * w;
  if (gr.k_w == gr.q_w + ww && gr.v_w == gr.k_w + ww) {
    RC(gemm_tn(W.d_qkv, 3 * w, L.ln1, w, M, 3 * w, w, gr.q_w, W.scratch, W.scratch_bytes, st));
  } else {
    RC(gemm_tn(W.d_qkv, 3 * w, L.ln1, w, M, w, w, gr.q_w, W.scratch, W.scratch_bytes, st));
    RC(gemm_tn(W.d_qkv + w, 3 * w, L.ln1, w, M, w, w, gr.k_w, W.scratch, W.scratch_bytes, st));
    RC(gemm_tn(W.d_qkv + 2 * w, 3 * w, L.ln1, w, M, w, w, gr.v_w, W.scratch, W.scratch_bytes, st));
  }
  {
    GemmEpi e; e.out_bf16 = W.d_ln; e.ld_ob = w;
    RC(gemm_nt(W.d_qkv, 3 * w, L.wqkv_t, 3 * w, M, w, 3 * w, e, st));
  }
  // LN1 (+ residual) over all rows: g = g2 + LNbwd(d_ln)
  RC(layernorm_bwd(W.d_ln, x_in, L.mean1, L.rstd1, p.ln1_w, g2, g, gb, gr.ln1_w, gr.ln1_b, M, w, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  return SMBV_OK;
}

static int layer_bwd(const smbv_layer_ptrs& p, const smbv_layer_ptrs& gr, LayerWs& L, MimWs& W, const float* x_in, float* g, bf16_t* gb,
                     float* g2, bf16_t* gb2, size_t rows, int w, int ff, int B, int S, int H, hipStream_t st) {
  const int M = (int)rows;
  // FC2
  RC(colsum_bf16(gb, w, M, w, gr.fc2_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(gemm_tn(gb, w, L.act, ff, M, w, ff, gr.fc2_w, W.scratch, W.scratch_bytes, st));
  {
    GemmEpi e; e.dgelu_pre = L.pre; e.ld_pre = ff; e.out_bf16 = W.d_act; e.ld_ob = ff;
    RC(gemm_nt(gb, w, L.fc2_t, w, M, ff, w, e, st));
  }
  // FC1
  RC(colsum_bf16(W.d_act, ff, M, ff, gr.fc1_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(gemm_tn(W.d_act, ff, L.ln2, w, M, ff, w, gr.fc1_w, W.scratch, W.scratch_bytes, st));
  {
    GemmEpi e; e.out_bf16 = W.d_ln; e.ld_ob = w;
    RC(gemm_nt(W.d_act, ff, L.fc1_t, ff, M, w, ff, e, st));
  }
  // LN2 (+ residual): g2 = g + LNbwd(d_ln)
  RC(layernorm_bwd(W.d_ln, L.x_mid, L.mean2, L.rstd2, p.ln2_w, g, g2, gb2, gr.ln2_w, gr.ln2_b, M, w, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  // attention output projection
  RC(colsum_bf16(gb2, w, M, w, gr.proj_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(gemm_tn(gb2, w, L.attn_o, w, M, w, w, gr.proj_w, W.scratch, W.scratch_bytes, st));
  {
    GemmEpi e; e.out_bf16 = W.d_o; e.ld_ob = w;
    RC(gemm_nt(gb2, w, L.proj_t, w, M, w, w, e, st));
  }
  // attention
  RC(attention_bwd(L.qkv, L.qkv + w, L.qkv + 2 * w, 3 * w, L.attn_o, W.d_o, w, L.lse, W.delta, W.d_qkv, W.d_qkv + w, W.d_qkv + 2 * w, 3 * w, B, S,
                   H, 64, 0.125f, 7, 0, 0, st));
  // QKV
  RC(colsum_bf16(W.d_qkv, 3 * w, M, w, gr.q_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(colsum_bf16(W.d_qkv + 2 * w, 3 * w, M, w, gr.v_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  const size_t ww = (size_t)w * w;
  if (gr.k_w == gr.q_w + ww && gr.v_w == gr.k_w + ww) {
    RC(gemm_tn(W.d_qkv, 3 * w, L.ln1, w, M, 3 * w, w, gr.q_w, W.scratch, W.scratch_bytes, st));
  } else {
    RC(gemm_tn(W.d_qkv, 3 * w, L.ln1, w, M, w, w, gr.q_w, W.scratch, W.scratch_bytes, st));
    RC(gemm_tn(W.d_qkv + w, 3 * w, L.ln1, w, M, w, w, gr.k_w, W.scratch, W.scratch_bytes, st));
    RC(gemm_tn(W.d_qkv + 2 * w, 3 * w, L.ln1, w, M, w, w, gr.v_w, W.scratch, W.scratch_bytes, st));
  }
  {
    GemmEpi e; e.out_bf16 = W.d_ln; e.ld_ob = w;
    RC(gemm_nt(W.d_qkv, 3 * w, L.wqkv_t, 3 * w, M, w, 3 * w, e, st));
  }
  // LN1 (+ residual): g = g2 + LNbwd(d_ln)
  RC(layernorm_bwd(W.d_ln, x_in, L.mean1, L.rstd1, p.ln1_w, g2, g, gb, gr.ln1_w, gr.ln1_b, M, w, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  return SMBV_OK;
}

static int mim_forward(const smbv_mim_cfg& c, const smbv_mim_tensors& p, const smbv_mim_io& io, void* ws, size_t ws_bytes, hipStream_t st) {
  RC(check_cfg(c));
  MimWs W;
  carve(c, ws, W);
  if (!ws || ws_bytes < W.total || ((uintptr_t)ws & 255)) { set_error("mim_forward: workspace too small or misaligned (%zu < %zu)", ws_bytes, W.total); return SMBV_ERR_WORKSPACE; }
  if (!io.pixel_values || !io.vis_idx || !io.mask_idx || !io.pos_enc || !io.pos_dec || !io.loss || !io.logits) { set_error("mim_forward: null io pointer"); return SMBV_ERR_ARG; }
  const int K = c.ts * c.ps * c.ps;
  const size_t R = (size_t)c.B * c.Nvis, Rd = (size_t)c.B * c.N, Rm = (size_t)c.B * c.Nmask;

  // bf16 operand copies (plain and transposed) of the fp32 master weights, which change every optimizer step: one launch
  {
    PrepList pl;
    pl.cast(p.pe_w, W.pe_w, c.d, K);
    pl.cast(p.e2d_w, W.e2d, c.dd, c.d);
    pl.tcast(p.e2d_w, W.e2d_t, c.dd, c.d, c.dd);
    pl.cast(p.head_w, W.head, K, c.dd);
    pl.tcast(p.head_w, W.head_t, K, c.dd, K);
    for (int l = 0; l < c.layers; l++) prep_layer(pl, p.enc[l], W.enc[l], c.d, c.ff);
    for (int l = 0; l < c.dlayers; l++) prep_layer(pl, p.dec[l], W.dec[l], c.dd, c.dff);
    if (pl.jobs.size() > 16 + 16 * (size_t)(c.layers + c.dlayers)) { set_error("mim_forward: prep table overflow"); return SMBV_ERR_WORKSPACE; }
    // pageable-source H2D: HIP stages the bytes before returning, so the vector may die at the end of this scope
    HIPRC(hipMemcpyAsync(W.prep_table, pl.jobs.data(), pl.jobs.size() * sizeof(PrepJob), hipMemcpyHostToDevice, st));
    RC(prep_weights(W.prep_table, (int)pl.jobs.size(), pl.tiles, st));
  }

  // patch embedding of the visible bricks + bias + sin-cos rows (spec :124-139, :179-192)
  RC(gather_bricks(io.pixel_values, io.vis_idx, (int)R, c.Nvis, W.a_patch, c.T, c.H, c.W, c.ts, c.ps, st));
  {
    GemmEpi e; e.bias = p.pe_b; e.rowadd = io.pos_enc; e.rowidx = io.vis_idx; e.ldr = c.d; e.out_f32 = W.x[0]; e.ld_of = c.d;
    RC(gemm_nt(W.a_patch, K, W.pe_w, K, (int)R, c.d, K, e, st));
  }
  for (int l = 0; l < c.layers; l++)
    RC(layer_fwd(p.enc[l], W.enc[l], W.x[l], W.x[l + 1], l == c.layers - 1 ? W.enc_out : nullptr, R, c.d, c.ff, c.B, c.Nvis, c.heads, c.ln_eps, W, 0, 0, st));
  // encoder_to_decoder + decoder input assembly: [visible + pos_vis, mask_token + pos_mask] (spec :801-815)
  {
    GemmEpi e; e.rowadd = io.pos_dec; e.rowidx = io.vis_idx; e.ldr = c.dd; e.out_f32 = W.xd[0]; e.ld_of = c.dd; e.grp_in = c.Nvis; e.grp_out = c.N;
    RC(gemm_nt(W.enc_out, c.d, W.e2d, c.d, (int)R, c.dd, c.d, e, st));
  }
  RC(fill_mask_rows(W.xd[0], p.mask_token, io.pos_dec, io.mask_idx, c.B, c.Nmask, c.Nvis, c.dd, st));
  for (int l = 0; l < c.dlayers; l++)
    RC(layer_fwd(p.dec[l], W.dec[l], W.xd[l], W.xd[l + 1], nullptr, Rd, c.dd, c.dff, c.B, c.N, c.dheads, c.ln_eps, W,
                 l == c.dlayers - 1 ? c.Nmask : 0, l == c.dlayers - 1 ? c.Nvis : 0, st));
  // last N_mask tokens -> decoder.norm -> head (spec :717-722)
  RC(layernorm_fwd(W.xd[c.dlayers], p.dnorm_w, p.dnorm_b, W.lnf, W.meanf, W.rstdf, (int)Rm, c.dd, c.dec_norm_eps, c.Nmask, c.N, c.Nvis, st));
  {
    GemmEpi e; e.bias = p.head_b; e.out_bf16 = io.logits; e.ld_ob = K;
    RC(gemm_nt(W.lnf, c.dd, W.head, c.dd, (int)Rm, K, c.dd, e, st));
  }
  // per-patch normalised targets + MSE (spec :822-897)
  RC(brick_stats(io.pixel_values, io.mask_idx, (int)Rm, c.Nmask, W.bmean, W.brinv, c.T, c.H, c.W, c.ts, c.ps, st));
  RC(mse_fwd(io.pixel_values, io.mask_idx, (int)Rm, c.Nmask, W.bmean, W.brinv, io.logits, W.row_partial, io.loss, c.T, c.H, c.W, c.ts, c.ps, st));
  return SMBV_OK;
}

static int mim_backward(const smbv_mim_cfg& c, const smbv_mim_tensors& p, const smbv_mim_tensors& gr, const smbv_mim_io& io, void* ws,
                        size_t ws_bytes, void** events, int n_events, hipStream_t st) {
  RC(check_cfg(c));
  MimWs W;
  carve(c, ws, W);
  if (!ws || ws_bytes < W.total || ((uintptr_t)ws & 255)) { set_error("mim_backward: workspace too small or misaligned"); return SMBV_ERR_WORKSPACE; }
  const int need_events = 3 + c.layers + c.dlayers;
  if (events && n_events < need_events) { set_error("mim_backward: need %d events, got %d", need_events, n_events); return SMBV_ERR_ARG; }
  int ev = 0;
  auto mark = [&]() -> int {
    if (events) HIPRC(hipEventRecord((hipEvent_t)events[ev], st));
    ev++;
    return SMBV_OK;
  };
  const int K = c.ts * c.ps * c.ps;
  const size_t R = (size_t)c.B * c.Nvis, Rd = (size_t)c.B * c.N, Rm = (size_t)c.B * c.Nmask;

  RC(mse_bwd(io.pixel_values, io.mask_idx, (int)Rm, c.Nmask, W.bmean, W.brinv, io.logits, W.dlogits, io.grad_scale, c.T, c.H, c.W, c.ts, c.ps, st));
  // head
  RC(colsum_bf16(W.dlogits, K, (int)Rm, K, gr.head_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(gemm_tn(W.dlogits, K, W.lnf, c.dd, (int)Rm, K, c.dd, gr.head_w, W.scratch, W.scratch_bytes, st));
  {
    GemmEpi e; e.out_bf16 = W.d_ln; e.ld_ob = c.dd;
    RC(gemm_nt(W.dlogits, K, W.head_t, K, (int)Rm, c.dd, K, e, st));
  }
  // decoder.norm backward scatters into the last N_mask rows of every volume; visible rows get no gradient from the head
  HIPRC(hipMemsetAsync(W.g0, 0, Rd * c.dd * sizeof(float), st));
  HIPRC(hipMemsetAsync(W.gb0, 0, Rd * c.dd * sizeof(bf16_t), st));
  RC(layernorm_bwd(W.d_ln, W.xd[c.dlayers], W.meanf, W.rstdf, p.dnorm_w, nullptr, W.g0, W.gb0, gr.dnorm_w, gr.dnorm_b, (int)Rm, c.dd, c.Nmask, c.N,
                   c.Nvis, W.scratch, W.scratch_bytes, st));
  RC(mark());
  for (int l = c.dlayers - 1; l >= 0; l--) {
    if (l == c.dlayers - 1)
      RC(layer_bwd_ranged(p.dec[l], gr.dec[l], W.dec[l], W, W.xd[l], W.g0, W.gb0, W.g1, W.gb1, Rd, c.dd, c.dff, c.B, c.N, c.dheads, c.Nmask, c.Nvis, st));
    else
      RC(layer_bwd(p.dec[l], gr.dec[l], W.dec[l], W, W.xd[l], W.g0, W.gb0, W.g1, W.gb1, Rd, c.dd, c.dff, c.B, c.N, c.dheads, st));
    RC(mark());
  }
  // decoder input: mask_token gradient = sum over masked rows; visible rows flow into encoder_to_decoder
  RC(colsum_f32(W.g0, c.dd, (int)Rm, c.dd, gr.mask_token, c.Nmask, c.N, c.Nvis, W.scratch, W.scratch_bytes, st));
  RC(gather_rows_bf16(W.gb0, W.d_vis, (int)R, c.dd, c.Nvis, c.N, 0, st));
  RC(gemm_tn(W.d_vis, c.dd, W.enc_out, c.d, (int)R, c.dd, c.d, gr.e2d_w, W.scratch, W.scratch_bytes, st));
  {
    GemmEpi e; e.out_f32 = W.g1; e.ld_of = c.d; e.out_bf16 = W.gb1; e.ld_ob = c.d;
    RC(gemm_nt(W.d_vis, c.dd, W.e2d_t, c.dd, (int)R, c.d, c.dd, e, st));
  }
  RC(mark());
  for (int l = c.layers - 1; l >= 0; l--) {
    RC(layer_bwd(p.enc[l], gr.enc[l], W.enc[l], W, W.x[l], W.g1, W.gb1, W.g0, W.gb0, R, c.d, c.ff, c.B, c.Nvis, c.heads, st));
    RC(mark());
  }
  // patch embedding (the volume needs no gradient)
  RC(colsum_bf16(W.gb1, c.d, (int)R, c.d, gr.pe_b, 0, 0, 0, W.scratch, W.scratch_bytes, st));
  RC(gemm_tn(W.gb1, c.d, W.a_patch, K, (int)R, c.d, K, gr.pe_w, W.scratch, W.scratch_bytes, st));
  RC(mark());
  return SMBV_OK;
}

}  // namespace smbv

extern "C" {

size_t smbv_mim_workspace_bytes(const smbv_mim_cfg* cfg) {
  if (!cfg || smbv::check_cfg(*cfg)) return 0;
  smbv::MimWs W;
  smbv::carve(*cfg, nullptr, W);
  return W.total;
}

int smbv_mim_forward(const smbv_mim_cfg* cfg, const smbv_mim_tensors* params, const smbv_mim_io* io, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!cfg || !params || !io) { smbv::set_error("smbv_mim_forward: null argument"); return SMBV_ERR_ARG; }
  return smbv::mim_forward(*cfg, *params, *io, ws, ws_bytes, (hipStream_t)stream);
}

int smbv_mim_backward(const smbv_mim_cfg* cfg, const smbv_mim_tensors* params, const smbv_mim_tensors* grads, const smbv_mim_io* io, void* ws,
                      size_t ws_bytes, void** events, int n_events, smbv_stream stream) {
  if (!cfg || !params || !grads || !io) { smbv::set_error("smbv_mim_backward: null argument"); return SMBV_ERR_ARG; }
  return smbv::mim_backward(*cfg, *params, *grads, *io, ws, ws_bytes, events, n_events, (hipStream_t)stream);
}

}  // extern "C"
