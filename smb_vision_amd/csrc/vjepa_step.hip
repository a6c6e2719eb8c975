// Host-side sequencing of the V-JEPA online-model step: VJEPA2Model.forward with context / target masks
// (src/models/vjepa/modeling_vjepa.py:1081-1163 -> encoder :507-539, predictor :699-746) and its autograd, over the kernels of
// this library.  The L1 loss against the EMA target encoder and the EMA update (src/run_vjepa.py:110-141) are separate entry
// points (smbv_l1_*, smbv_ema_update) driven by the trainer mirror, exactly where the reference computes them.
//
// The encoder runs over ALL N tokens (no masking before it, :507-539).  The predictor sequence is held in position-sorted order
// from the start (the reference concatenates [context, target] and then gathers by argsort, :718-719): context embeddings are
// scattered to their slots, target slots are filled with the mask token, and RoPE takes the sorted position ids.
// Predictor heads (head_dim 32 at the default 384/12) use the native 32-wide attention build; widths other than 32 / 64 run zero-padded in 64-wide slots (block.h).
#include "block.h"
#include "../../include/smbv.h"

namespace smbv {

struct VjepaWs {
  bf16_t *pe_w, *pemb, *pemb_t, *pproj, *pproj_t;
  LayerWs enc[SMBV_MAX_LAYERS], pred[SMBV_MAX_LAYERS];
  bf16_t* a_patch;
  float* x[SMBV_MAX_LAYERS + 1];    // encoder residual stream [B*N, d]
  float* xp[SMBV_MAX_LAYERS + 1];   // predictor residual stream [B*Sp, pd]
  bf16_t *enc_ln, *ctx_bf, *pred_ln, *tgt_bf;
  float *enc_mean, *enc_rstd, *pred_mean, *pred_rstd, *ctx_emb;
  // backward
  float *g0, *g1, *gtmp;
  bf16_t *dpred_bf, *d_tgt, *d_lnfull, *d_ctx, *d_encctx;
  BwdScratch sc;
  WgradPipe pipe;   // rings of the deferred weight-gradient launches (block.h)
  size_t total;
};

static BlockDims venc_dims(const smbv_vjepa_cfg& c, const smbv_vjepa_io* io) {
  BlockDims d; d.rows = (size_t)c.B * c.N; d.w = c.d; d.ff = c.ff; d.B = c.B; d.S = c.N; d.H = c.heads; d.Dh = c.d / c.heads; d.eps = c.ln_eps;
  d.o_resid = c.attn_o_resid; d.rope_grid = c.W / c.ps; d.pos = io ? io->enc_pos : nullptr; d.rope_table = io ? io->rope_enc : nullptr;
  d.rope_rows = std::max(c.T / c.ts, std::max(c.H / c.ps, c.W / c.ps));   // every frame / height / width id is below it (include/smbv.h: the table covers them)
  return d;
}
static inline int vmasks(const smbv_vjepa_cfg& c) { return c.n_masks > 1 ? c.n_masks : 1; }
// rows of every volume whose last-block output is produced (0 = all: the last block is not ranged)
static inline int vneed(const smbv_vjepa_cfg& c) { return (c.enc_needed > 0 && c.enc_needed < c.N) ? c.enc_needed : 0; }
// rows of every predictor sequence the last predictor block computes (0 = all): the target slots when the caller put them first
static inline int vpneed(const smbv_vjepa_cfg& c) { const int dh = c.pd / c.pheads; return (c.pred_tgt_first && c.Nctx > 0 && (dh == 64 || dh == 32)) ? c.Ntgt : 0; }
static inline int vpb(const smbv_vjepa_cfg& c) { return c.B * vmasks(c); }  // predictor sequences: mask pair m, volume b -> m * B + b
static BlockDims vpred_dims(const smbv_vjepa_cfg& c, const smbv_vjepa_io* io) {
  const int Sp = c.Nctx + c.Ntgt;
  BlockDims d; d.rows = (size_t)vpb(c) * Sp; d.w = c.pd; d.ff = c.pff; d.B = vpb(c); d.S = Sp; d.H = c.pheads; d.Dh = c.pd / c.pheads; d.eps = c.ln_eps;
  d.o_resid = c.attn_o_resid; d.rope_grid = c.W / c.ps; d.pos = io ? io->pred_pos : nullptr; d.rope_table = io ? io->rope_pred : nullptr;
  d.rope_rows = std::max(c.T / c.ts, std::max(c.H / c.ps, c.W / c.ps));
  return d;
}
static BlockPtrs vblock(const smbv_block_ptrs& p) {
  BlockPtrs b;
  b.ln1_w = p.ln1_w; b.ln1_b = p.ln1_b; b.q_w = p.q_w; b.k_w = p.k_w; b.v_w = p.v_w; b.q_b = p.q_b; b.k_b = p.k_b; b.v_b = p.v_b;
  b.proj_w = p.proj_w; b.proj_b = p.proj_b; b.ln2_w = p.ln2_w; b.ln2_b = p.ln2_b; b.fc1_w = p.fc1_w; b.fc1_b = p.fc1_b; b.fc2_w = p.fc2_w; b.fc2_b = p.fc2_b;
  return b;
}

static int vcheck(const smbv_vjepa_cfg& c) {
  if (c.struct_bytes != (int)sizeof(smbv_vjepa_cfg)) { set_error("vjepa: cfg.struct_bytes = %d, this library's smbv_vjepa_cfg is %d bytes (binding built against another ABI: check smbv_version())", c.struct_bytes, (int)sizeof(smbv_vjepa_cfg)); return SMBV_ERR_ARG; }
  if (c.n_masks < 0 || c.n_masks > 64) { set_error("vjepa: n_masks %d out of range", c.n_masks); return SMBV_ERR_SHAPE; }
  if (c.enc_needed < 0 || c.enc_needed > c.N || (c.enc_needed > 0 && c.enc_needed < c.N && c.n_masks > 1)) { set_error("vjepa: enc_needed %d out of range (0..N; with one mask pair only)", c.enc_needed); return SMBV_ERR_SHAPE; }
  if (c.B <= 0 || c.layers <= 0 || c.players < 0 || c.layers > SMBV_MAX_LAYERS || c.players > SMBV_MAX_LAYERS) { set_error("vjepa: bad batch/layer counts"); return SMBV_ERR_SHAPE; }
  if (c.ps <= 0 || c.ts <= 0 || c.T % c.ts || c.H % c.ps || c.W % c.ps || (c.ps & 3) || c.H != c.W) { set_error("vjepa: volume %dx%dx%d vs tubelet %d / patch %d (square crops only)", c.T, c.H, c.W, c.ts, c.ps); return SMBV_ERR_SHAPE; }
  const int N = (c.T / c.ts) * (c.H / c.ps) * (c.W / c.ps);
  if (c.N != N || c.Nctx <= 0 || c.Ntgt <= 0 || c.Nctx > N || c.Ntgt > N) { set_error("vjepa: token counts N=%d Nctx=%d Ntgt=%d inconsistent (%d)", c.N, c.Nctx, c.Ntgt, N); return SMBV_ERR_SHAPE; }
  if (c.d % c.heads || c.pd % c.pheads) { set_error("vjepa: width not divisible by heads"); return SMBV_ERR_SHAPE; }
  const int dh = c.d / c.heads, pdh = c.pd / c.pheads;
  if (dh > 64 || pdh > 64 || (dh & 7) || (pdh & 7)) { set_error("vjepa: head_dim must be a multiple of 8 and <= 64 (got %d / %d)", dh, pdh); return SMBV_ERR_SHAPE; }
  if ((c.d & 7) || (c.pd & 7) || (c.ff & 7) || (c.pff & 7) || ((c.ts * c.ps * c.ps) & 7)) { set_error("vjepa: widths must be multiples of 8"); return SMBV_ERR_SHAPE; }
  return SMBV_OK;
}

static void vcarve(const smbv_vjepa_cfg& c, void* base, VjepaWs& w) {
  Carver cv{(char*)base};
  const int K = c.ts * c.ps * c.ps;
  const BlockDims de = venc_dims(c, nullptr), dp = vpred_dims(c, nullptr);
  const size_t Re = de.rows, Rp = dp.rows, Rc = (size_t)vpb(c) * c.Nctx, Rt = (size_t)vpb(c) * c.Ntgt;
  w.pe_w = cv.take<bf16_t>((size_t)c.d * K);
  w.pemb = cv.take<bf16_t>((size_t)c.pd * c.d);
  w.pemb_t = cv.take<bf16_t>((size_t)c.pd * c.d);
  w.pproj = cv.take<bf16_t>((size_t)c.pd * c.d);
  w.pproj_t = cv.take<bf16_t>((size_t)c.pd * c.d);
  for (int l = 0; l < c.layers; l++) carve_layer(cv, w.enc[l], de);
  for (int l = 0; l < c.players; l++) carve_layer(cv, w.pred[l], dp);
  w.a_patch = cv.take<bf16_t>(Re * K);
  for (int l = 0; l <= c.layers; l++) w.x[l] = cv.take<float>(Re * c.d);
  for (int l = 0; l <= c.players; l++) w.xp[l] = cv.take<float>(Rp * c.pd);
  w.enc_ln = cv.take<bf16_t>(Re * c.d);
  w.enc_mean = cv.take<float>(Re);
  w.enc_rstd = cv.take<float>(Re);
  w.ctx_bf = cv.take<bf16_t>(Rc * c.d);
  w.ctx_emb = cv.take<float>(Rc * c.pd);
  w.pred_ln = cv.take<bf16_t>(Rp * c.pd);
  w.pred_mean = cv.take<float>(Rp);
  w.pred_rstd = cv.take<float>(Rp);
  w.tgt_bf = cv.take<bf16_t>(Rt * c.pd);
  const size_t gw = std::max(Re * c.d, Rp * c.pd);
  w.g0 = cv.take<float>(gw);
  w.g1 = cv.take<float>(gw);
  w.gtmp = cv.take<float>(Rt * c.pd);
  w.dpred_bf = cv.take<bf16_t>(Rt * c.d);
  w.d_tgt = cv.take<bf16_t>(Rt * c.pd);
  w.d_lnfull = cv.take<bf16_t>(gw);
  w.d_ctx = cv.take<bf16_t>(Rc * c.pd);
  w.d_encctx = cv.take<bf16_t>(Rc * c.d);
  BwdNeeds nd;
  bwd_scratch_requirements(de, nd);
  bwd_scratch_requirements(dp, nd);
  if (vneed(c)) {  // the ranged last block runs its MLP / projection products on B * enc_needed rows: their slab counts differ from the full-size ones
    BlockDims dr = de;
    dr.rows = (size_t)c.B * vneed(c);
    bwd_scratch_requirements(dr, nd);
  }
  if (vpneed(c)) {
    BlockDims dr = dp;
    dr.rows = (size_t)vpb(c) * vpneed(c);
    bwd_scratch_requirements(dr, nd);
  }
  auto tn = [&](size_t M, int n1, int n2) { nd.scratch = std::max(nd.scratch, gemm_tn_workspace_bytes((int)M, n1, n2)); };
  tn(Re, c.d, K); tn(Rc, c.pd, c.d); tn(Rt, c.d, c.pd);
  nd.scratch = std::max(nd.scratch, colsum_workspace_bytes((int)std::max(Re, Rp), std::max(std::max(c.d, c.pd), K)));
  carve_bwd_scratch(cv, w.sc, nd);
  carve_wgrad_pipe(cv, w.pipe, gw, nd);
  w.sc.gath0 = nullptr;
  w.sc.gath1 = nullptr;
  {  // gathered copies of the ranged rows (block.hip: contiguous without them only for one sequence)
    size_t ge = 0;
    if (vneed(c) && c.B > 1) ge = std::max(ge, (size_t)c.B * vneed(c) * c.d);
    if (vpneed(c) && vpb(c) > 1) ge = std::max(ge, (size_t)vpb(c) * vpneed(c) * c.pd);
    if (ge) { w.sc.gath0 = cv.take<bf16_t>(ge); w.sc.gath1 = cv.take<bf16_t>(ge); }
  }
  w.total = (cv.off + 255) & ~(size_t)255;
}

static int vjepa_forward(const smbv_vjepa_cfg& c, const smbv_vjepa_tensors& p, const smbv_vjepa_io& io, void* ws, size_t ws_bytes, hipStream_t st) {
  RC(vcheck(c));
  VjepaWs W;
  vcarve(c, ws, W);
  if (!ws || ws_bytes < W.total || ((uintptr_t)ws & 255)) { set_error("vjepa_forward: workspace too small or misaligned (%zu < %zu)", ws_bytes, W.total); return SMBV_ERR_WORKSPACE; }
  const bool with_pred = io.pred_out != nullptr;
  if (!io.pixel_values || !io.all_idx || !io.rope_enc || !io.enc_out) { set_error("vjepa_forward: null io pointer"); return SMBV_ERR_ARG; }
  if (with_pred && (!io.ctx_idx || !io.tgt_idx || !io.pred_pos || !io.ctx_slot || !io.tgt_slot || !io.rope_pred || c.players <= 0)) { set_error("vjepa_forward: predictor inputs missing"); return SMBV_ERR_ARG; }
  const int K = c.ts * c.ps * c.ps;
  const BlockDims de = venc_dims(c, &io), dp = vpred_dims(c, &io);
  const size_t Re = de.rows, Rp = dp.rows, Rc = (size_t)vpb(c) * c.Nctx, Rt = (size_t)vpb(c) * c.Ntgt;
  const int Sp = c.Nctx + c.Ntgt;
  SideStream prep_side;
  int prep_join_layer = -1;
  {
    PrepList pl;
    pl.cast(p.pe_w, W.pe_w, c.d, K);
    int first_block_end = 0;   // jobs [0, first_block_end): the patch-embedding weight and the first block's copies
    for (int l = 0; l < c.layers; l++) {
      RC(prep_layer(pl, vblock(p.enc[l]), W.enc[l], de, st));
      if (l == 0) first_block_end = (int)pl.jobs.size();
    }
    if (with_pred) {
      pl.both(p.pemb_w, W.pemb, c.d, W.pemb_t, c.pd, c.pd, c.d);
      pl.both(p.pproj_w, W.pproj, c.pd, W.pproj_t, c.d, c.d, c.pd);
      for (int l = 0; l < c.players; l++) RC(prep_layer(pl, vblock(p.pred[l]), W.pred[l], dp, st));
    }
    // as in mim_step.hip: what the first kernels read (patch-embedding weight + the first block's copies) on `st`, the rest on the side stream beside the
    // brick gather, the patch-embedding product and the first block, joined in front of the second block
    const int njobs = (int)pl.jobs.size(), first = std::min(njobs, first_block_end);
    const bool split = njobs > first && !de.padded() && !(with_pred && dp.padded());
    RC(prep_weights(pl.jobs.data(), split ? first : njobs, st));
    if (split) {
      RC(prep_side.begin(st));
      RC(prep_weights(pl.jobs.data() + first, njobs - first, prep_side.st));
      prep_join_layer = 1;
    }
  }
  // encoder over all N tokens: Conv3d patch embedding (+bias, no position table: RoPE instead), L blocks, final LayerNorm
  RC(gather_bricks(io.pixel_values, io.all_idx, (int)Re, c.N, W.a_patch, c.T, c.H, c.W, c.ts, c.ps, st));
  {
    GemmEpi e; e.bias = p.pe_b; e.out_f32 = W.x[0]; e.ld_of = c.d;
    RC(gemm_nt(W.a_patch, K, W.pe_w, K, (int)Re, c.d, K, e, st));
  }
  // enc_needed: the caller put the tokens whose encoder output it consumes first; the last block then works on those queries only and the final
  // LayerNorm / enc_out / W.enc_ln are compact [B * need, d]
  const int need = vneed(c);
  if (need && de.padded()) { set_error("vjepa_forward: enc_needed with head-padded encoder widths is not supported"); return SMBV_ERR_SHAPE; }
  for (int l = 0; l < c.layers; l++) {
    if (l == prep_join_layer) RC(prep_side.end(st));
    RC(layer_fwd(vblock(p.enc[l]), W.enc[l], de, W.sc, W.x[l], W.x[l + 1], nullptr, l == c.layers - 1 ? need : 0, 0, st));
  }
  if (prep_join_layer >= c.layers) RC(prep_side.end(st));
  if (need) RC(layernorm_fwd(W.x[c.layers], p.enc_ln_w, p.enc_ln_b, W.enc_ln, W.enc_mean, W.enc_rstd, c.B * need, c.d, c.ln_eps, need, c.N, 0, st, io.enc_out));
  else RC(layernorm_fwd(W.x[c.layers], p.enc_ln_w, p.enc_ln_b, W.enc_ln, W.enc_mean, W.enc_rstd, (int)Re, c.d, c.ln_eps, 0, 0, 0, st, io.enc_out));
  if (!with_pred) return SMBV_OK;
  // predictor: context rows -> Linear(d -> pd) -> their slots of the position-sorted sequence; target slots <- mask token
  RC(gather_rows_idx_bf16(W.enc_ln, W.ctx_bf, io.ctx_idx, (long)Rc, (int)Rc, 0, c.d, st));  // ctx_idx holds absolute rows of enc_out
  {
    GemmEpi e; e.bias = p.pemb_b; e.out_f32 = W.ctx_emb; e.ld_of = c.pd;
    RC(gemm_nt(W.ctx_bf, c.d, W.pemb, c.d, (int)Rc, c.pd, c.d, e, st));
  }
  RC(scatter_rows_idx_f32(W.ctx_emb, W.xp[0], io.ctx_slot, (long)Rc, c.Nctx, Sp, c.pd, st));
  RC(fill_rows_idx_f32(W.xp[0], p.mask_token, io.tgt_slot, (long)Rt, c.Ntgt, Sp, c.pd, st));
  // pred_tgt_first: the target tokens sit in slots 0 .. Ntgt-1; the last block computes those rows only, the final LayerNorm output is compact [P * Ntgt, pd]
  const int pneed = vpneed(c);
  for (int l = 0; l < c.players; l++) RC(layer_fwd(vblock(p.pred[l]), W.pred[l], dp, W.sc, W.xp[l], W.xp[l + 1], nullptr, l == c.players - 1 ? pneed : 0, 0, st));
  if (pneed) RC(layernorm_fwd(W.xp[c.players], p.pred_ln_w, p.pred_ln_b, W.pred_ln, W.pred_mean, W.pred_rstd, vpb(c) * pneed, c.pd, c.ln_eps, pneed, Sp, 0, st));
  else RC(layernorm_fwd(W.xp[c.players], p.pred_ln_w, p.pred_ln_b, W.pred_ln, W.pred_mean, W.pred_rstd, (int)Rp, c.pd, c.ln_eps, 0, 0, 0, st));
  // unsort + keep the target tokens (:737-738) == gather the target slots; then proj (pd -> d)
  RC(gather_rows_idx_bf16(W.pred_ln, W.tgt_bf, io.tgt_slot, (long)Rt, c.Ntgt, pneed ? pneed : Sp, c.pd, st));
  {
    GemmEpi e; e.bias = p.pproj_b; e.out_f32 = io.pred_out; e.ld_of = c.d;
    RC(gemm_nt(W.tgt_bf, c.pd, W.pproj, c.pd, (int)Rt, c.d, c.pd, e, st));
  }
  return SMBV_OK;
}

static int vjepa_backward(const smbv_vjepa_cfg& c, const smbv_vjepa_tensors& p, const smbv_vjepa_tensors& gr, const smbv_vjepa_io& io, void* ws,
                          size_t ws_bytes, void** events, int n_events, hipStream_t st) {
  RC(vcheck(c));
  const int need_events = 4 + c.layers + c.players;
  if (events && n_events < need_events) { set_error("vjepa_backward: need %d events, got %d", need_events, n_events); return SMBV_ERR_ARG; }
  int ev = 0;
  VjepaWs W;
  vcarve(c, ws, W);
  WgradPipe& pipe = W.pipe;
  RC(pipe.begin());
  auto next_event = [&]() -> hipEvent_t { return events ? (hipEvent_t)events[ev++] : (ev++, (hipEvent_t) nullptr); };
  // a gradient group produced on `st` is final: the data-parallel engine may start averaging it (the blocks' own groups: layer_bwd, on the side stream)
  auto mark = [&]() -> int { return pipe.mark_inline(next_event(), st); };
  if (!ws || ws_bytes < W.total || ((uintptr_t)ws & 255)) { set_error("vjepa_backward: workspace too small or misaligned"); return SMBV_ERR_WORKSPACE; }
  const bool with_pred = io.d_pred != nullptr;
  if (!with_pred && !io.d_enc) { set_error("vjepa_backward: neither d_pred nor d_enc"); return SMBV_ERR_ARG; }
  if (with_pred && (!io.tgt_slot || !io.ctx_slot || !io.ctx_idx)) { set_error("vjepa_backward: null io pointer"); return SMBV_ERR_ARG; }
  if (io.d_enc && vneed(c)) { set_error("vjepa_backward: d_enc with enc_needed (the full encoder output was not produced)"); return SMBV_ERR_ARG; }
  const int K = c.ts * c.ps * c.ps;
  const BlockDims de = venc_dims(c, &io), dp = vpred_dims(c, &io);
  const size_t Re = de.rows, Rp = dp.rows, Rc = (size_t)vpb(c) * c.Nctx, Rt = (size_t)vpb(c) * c.Ntgt;
  const int Sp = c.Nctx + c.Ntgt;
  if (with_pred) {
  // predictor.proj
  RC(cast_f32_bf16(io.d_pred, W.dpred_bf, Rt * c.d, st));
  {
    TnColsum cs; cs.dst[0] = gr.pproj_b; cs.seg = c.d;
    RC(gemm_tn(W.dpred_bf, c.d, W.tgt_bf, c.pd, (int)Rt, c.d, c.pd, gr.pproj_w, W.sc.scratch, W.sc.scratch_bytes, st, &cs));
  }
  {
    GemmEpi e; e.out_bf16 = W.d_tgt; e.ld_ob = c.pd;
    RC(gemm_nt(W.dpred_bf, c.d, W.pproj_t, c.d, (int)Rt, c.pd, c.d, e, st));
  }
  // predictor.layernorm: only the target slots receive gradient
  const int pneed = vpneed(c);
  HIPRC(hipMemsetAsync(W.d_lnfull, 0, Rp * c.pd * sizeof(bf16_t), st));
  RC(scatter_rows_idx_bf16(W.d_tgt, W.d_lnfull, io.tgt_slot, (long)Rt, c.Ntgt, pneed ? pneed : Sp, c.pd, st));
  if (pneed) RC(layernorm_bwd(W.d_lnfull, W.xp[c.players], W.pred_mean, W.pred_rstd, p.pred_ln_w, nullptr, W.g0, pipe.cur(), gr.pred_ln_w, gr.pred_ln_b, vpb(c) * pneed, c.pd,
                              pneed, Sp, 0, W.sc.scratch, W.sc.scratch_bytes, st));   // compact dy / statistics; gradients land at the slots' rows of the full layout
  else RC(layernorm_bwd(W.d_lnfull, W.xp[c.players], W.pred_mean, W.pred_rstd, p.pred_ln_w, nullptr, W.g0, pipe.cur(), gr.pred_ln_w, gr.pred_ln_b, (int)Rp, c.pd, 0,
                        0, 0, W.sc.scratch, W.sc.scratch_bytes, st));
  RC(mark());  // [0] predictor.proj + predictor.layernorm
  for (int l = c.players - 1; l >= 0; l--) {
    RC(layer_bwd(vblock(p.pred[l]), vblock(gr.pred[l]), W.pred[l], dp, W.sc, W.xp[l], W.g0, W.g1, pipe, l == c.players - 1 ? pneed : 0, 0, next_event(), st));   // [1 .. players] predictor layers, last first
  }
  // predictor input: mask-token gradient = sum over the target slots; context slots flow into predictor_embeddings
  RC(gather_rows_idx_f32(W.g0, W.gtmp, io.tgt_slot, (long)Rt, c.Ntgt, Sp, c.pd, st));
  RC(colsum_f32(W.gtmp, c.pd, (int)Rt, c.pd, gr.mask_token, 0, 0, 0, W.sc.scratch, W.sc.scratch_bytes, st));
  RC(gather_rows_idx_bf16(pipe.cur(), W.d_ctx, io.ctx_slot, (long)Rc, c.Nctx, Sp, c.pd, st));
  {
    TnColsum cs; cs.dst[0] = gr.pemb_b; cs.seg = c.pd;
    RC(gemm_tn(W.d_ctx, c.pd, W.ctx_bf, c.d, (int)Rc, c.pd, c.d, gr.pemb_w, W.sc.scratch, W.sc.scratch_bytes, st, &cs));
  }
  {
    GemmEpi e; e.out_bf16 = W.d_encctx; e.ld_ob = c.d;
    RC(gemm_nt(W.d_ctx, c.pd, W.pemb_t, c.pd, (int)Rc, c.d, c.pd, e, st));
  }
  RC(mark());  // [players+1] mask tokens + predictor_embeddings
  } else {
    for (int i = 0; i < c.players + 2; i++) RC(mark());  // the predictor's gradient groups: untouched by this call (the caller zeroed them)
  }
  // encoder.layernorm: only the context tokens receive gradient
  HIPRC(hipMemsetAsync(W.d_lnfull, 0, Re * c.d * sizeof(bf16_t), st));
  // mask pair 0 scatters, the others add on top (apply_masks' autograd sums the pairs that read the same encoder row); rows are distinct
  // inside a pair, the launches are stream-ordered: no atomics, reproducible
  if (with_pred) {
    const long per = (long)c.B * c.Nctx;
    RC(scatter_rows_idx_bf16(W.d_encctx, W.d_lnfull, io.ctx_idx, per, (int)per, 0, c.d, st));
    for (int m = 1; m < vmasks(c); m++) RC(scatter_add_rows_idx_bf16(W.d_encctx + (size_t)m * per * c.d, W.d_lnfull, io.ctx_idx + (size_t)m * per, per, c.d, st));
  }
  // gradient that reached the encoder output directly (last_hidden_state, masked / target_hidden_state): added on top, rounded to the bf16 the
  // LayerNorm backward takes its dy in
  if (io.d_enc) RC(add_f32_into_bf16(W.d_lnfull, io.d_enc, Re * c.d, st));
  const int need = vneed(c);
  if (need) RC(layernorm_bwd(W.d_lnfull, W.x[c.layers], W.enc_mean, W.enc_rstd, p.enc_ln_w, nullptr, W.g0, pipe.cur(), gr.enc_ln_w, gr.enc_ln_b, c.B * need, c.d, need, c.N, 0,
                             W.sc.scratch, W.sc.scratch_bytes, st));   // compact dy / statistics, gradients written at the tokens' rows of the full layout
  else RC(layernorm_bwd(W.d_lnfull, W.x[c.layers], W.enc_mean, W.enc_rstd, p.enc_ln_w, nullptr, W.g0, pipe.cur(), gr.enc_ln_w, gr.enc_ln_b, (int)Re, c.d, 0, 0, 0,
                        W.sc.scratch, W.sc.scratch_bytes, st));
  RC(mark());  // [players+2] encoder.layernorm
  for (int l = c.layers - 1; l >= 0; l--) {
    pipe.last_block = l == 0;   // its weight-gradient launch is the tail of the backward: token halves on twice the CUs (gemm_tn_grouped)
    const int rcl = layer_bwd(vblock(p.enc[l]), vblock(gr.enc[l]), W.enc[l], de, W.sc, W.x[l], W.g0, W.g1, pipe, l == c.layers - 1 ? need : 0, 0, next_event(), st);   // [players+3 ..] encoder layers, last first
    pipe.last_block = false;
    RC(rcl);
  }
  // patch embedding
  {
    TnColsum cs; cs.dst[0] = gr.pe_b; cs.seg = c.d;
    RC(gemm_tn(pipe.cur(), c.d, W.a_patch, K, (int)Re, c.d, K, gr.pe_w, W.sc.scratch, W.sc.scratch_bytes, st, &cs));
  }
  RC(mark());  // [last] patch embedding
  RC(pipe.drain(st));   // the caller's stream is ordered behind every deferred launch
  return SMBV_OK;
}

}  // namespace smbv

extern "C" {

size_t smbv_vjepa_workspace_bytes(const smbv_vjepa_cfg* cfg) {
  if (!cfg || smbv::vcheck(*cfg)) return 0;
  smbv::VjepaWs W;
  smbv::vcarve(*cfg, nullptr, W);
  return W.total;
}
int smbv_vjepa_forward(const smbv_vjepa_cfg* cfg, const smbv_vjepa_tensors* params, const smbv_vjepa_io* io, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!cfg || !params || !io) { smbv::set_error("smbv_vjepa_forward: null argument"); return SMBV_ERR_ARG; }
  SMBV_CHECK_DEVICE_STATE();
  return smbv::vjepa_forward(*cfg, *params, *io, ws, ws_bytes, (hipStream_t)stream);
}
int smbv_vjepa_backward(const smbv_vjepa_cfg* cfg, const smbv_vjepa_tensors* params, const smbv_vjepa_tensors* grads, const smbv_vjepa_io* io, void* ws,
                        size_t ws_bytes, void** events, int n_events, smbv_stream stream) {
  if (!cfg || !params || !grads || !io) { smbv::set_error("smbv_vjepa_backward: null argument"); return SMBV_ERR_ARG; }
  SMBV_CHECK_DEVICE_STATE();
  return smbv::vjepa_backward(*cfg, *params, *grads, *io, ws, ws_bytes, events, n_events, (hipStream_t)stream);
}

}  // extern "C"
