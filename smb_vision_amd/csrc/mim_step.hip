// Host-side sequencing of the whole MIM train step (VideoMAEForPreTraining.forward, spec
// src/models/videomae/modeling_videomae.py:753-908, and its autograd) over the kernels of this library.
// One C call enqueues the entire forward (or backward) on the caller's stream; nothing here synchronises or allocates.
//
// dtype policy (mirrors torch.autocast(bf16) as HF Trainer applies it): residual stream, LayerNorm
// statistics, softmax, MSE, all parameter gradients in fp32; GEMM / attention operands in bf16 with fp32 accumulation.
#include "block.h"
#include "../../include/smbv.h"

namespace smbv {

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
  float *g0, *g1;
  bf16_t *dlogits, *d_vis;
  BwdScratch sc;        // block-backward scratch (d_ln, d_o, gathers, delta, reduction scratch)
  WgradPipe pipe;       // bf16 residual-gradient ring, d_act / d_qkv / LayerNorm-partial rings of the deferred weight-gradient launches (block.h)
  size_t total;
};

static BlockDims enc_dims(const smbv_mim_cfg& c) {
  BlockDims d; d.rows = (size_t)c.B * c.Nvis; d.w = c.d; d.ff = c.ff; d.B = c.B; d.S = c.Nvis; d.H = c.heads; d.Dh = c.d / c.heads; d.eps = c.ln_eps; d.attn_fp8 = c.attn_fp8; d.o_resid = c.attn_o_resid;
  return d;
}
static BlockDims dec_dims(const smbv_mim_cfg& c) {
  BlockDims d; d.rows = (size_t)c.B * c.N; d.w = c.dd; d.ff = c.dff; d.B = c.B; d.S = c.N; d.H = c.dheads; d.Dh = c.dd / c.dheads; d.eps = c.ln_eps; d.attn_fp8 = c.attn_fp8; d.o_resid = c.attn_o_resid;
  return d;
}
static BlockPtrs to_block(const smbv_layer_ptrs& p) {
  BlockPtrs b;
  b.ln1_w = p.ln1_w; b.ln1_b = p.ln1_b; b.q_w = p.q_w; b.k_w = p.k_w; b.v_w = p.v_w; b.q_b = p.q_b; b.k_b = nullptr; b.v_b = p.v_b;
  b.proj_w = p.proj_w; b.proj_b = p.proj_b; b.ln2_w = p.ln2_w; b.ln2_b = p.ln2_b; b.fc1_w = p.fc1_w; b.fc1_b = p.fc1_b; b.fc2_w = p.fc2_w; b.fc2_b = p.fc2_b;
  return b;
}

static int check_cfg(const smbv_mim_cfg& c) {
  if (c.struct_bytes != (int)sizeof(smbv_mim_cfg)) { set_error("mim: cfg.struct_bytes = %d, this library's smbv_mim_cfg is %d bytes (binding built against another ABI: check smbv_version())", c.struct_bytes, (int)sizeof(smbv_mim_cfg)); return SMBV_ERR_ARG; }
  if (c.B <= 0 || c.layers <= 0 || c.dlayers <= 0 || c.layers > SMBV_MAX_LAYERS || c.dlayers > SMBV_MAX_LAYERS) { set_error("mim: bad batch/layer counts"); return SMBV_ERR_SHAPE; }
  if (c.ps <= 0 || c.ts <= 0 || c.T % c.ts || c.H % c.ps || c.W % c.ps || (c.ps & 3)) { set_error("mim: volume %dx%dx%d not divisible by tubelet %d / patch %d", c.T, c.H, c.W, c.ts, c.ps); return SMBV_ERR_SHAPE; }
  const int N = (c.T / c.ts) * (c.H / c.ps) * (c.W / c.ps);
  if (c.N != N || c.Nvis <= 0 || c.Nmask <= 0 || c.Nvis + c.Nmask != N) { set_error("mim: token counts N=%d Nvis=%d Nmask=%d inconsistent with the volume (%d)", c.N, c.Nvis, c.Nmask, N); return SMBV_ERR_SHAPE; }
  if (c.heads <= 0 || c.dheads <= 0 || c.d % c.heads || c.dd % c.dheads) { set_error("mim: width not divisible by heads (d=%d/%d, dd=%d/%d)", c.d, c.heads, c.dd, c.dheads); return SMBV_ERR_SHAPE; }
  // head widths 64 and 32 have native attention builds; other multiples of 8 below 64 run zero-padded in 64-wide slots (block.h: exact, slower)
  const int dh = c.d / c.heads, ddh = c.dd / c.dheads;
  if (dh > 64 || ddh > 64 || (dh & 7) || (ddh & 7)) { set_error("mim: head_dim must be a multiple of 8 and <= 64 (d=%d/%d, dd=%d/%d)", c.d, c.heads, c.dd, c.dheads); return SMBV_ERR_SHAPE; }
  if (c.attn_fp8 && (dh != 64 || ddh != 64)) { set_error("mim: the fp8 attention path is built for head_dim 64 (d=%d/%d, dd=%d/%d)", c.d, c.heads, c.dd, c.dheads); return SMBV_ERR_SHAPE; }
  if ((c.d & 7) || (c.dd & 7) || (c.ff & 7) || (c.dff & 7) || ((c.ts * c.ps * c.ps) & 7)) { set_error("mim: widths must be multiples of 8"); return SMBV_ERR_SHAPE; }
  return SMBV_OK;
}

static void carve(const smbv_mim_cfg& c, void* base, MimWs& w) {
  Carver cv{(char*)base};
  const int K = c.ts * c.ps * c.ps;
  const size_t R = (size_t)c.B * c.Nvis, Rd = (size_t)c.B * c.N, Rm = (size_t)c.B * c.Nmask;
  const BlockDims de = enc_dims(c), dd = dec_dims(c);
  w.pe_w = cv.take<bf16_t>((size_t)c.d * K);
  w.e2d = cv.take<bf16_t>((size_t)c.dd * c.d);
  w.e2d_t = cv.take<bf16_t>((size_t)c.dd * c.d);
  w.head = cv.take<bf16_t>((size_t)K * c.dd);
  w.head_t = cv.take<bf16_t>((size_t)K * c.dd);
  // recompute (gradient checkpointing, spec :459-465): the blocks of a stack share TWO activation sets (even / odd layers); only their inputs x[l] are
  // kept.  Two, not one, so that a block's deferred weight-gradient launch - which reads the block's activations - runs beside the NEXT block's
  // re-run forward and backward (that block writes the other set) instead of in front of them: -7 % of the ViT-large step, +1 set of memory.
  for (int l = 0; l < c.layers; l++) carve_layer(cv, w.enc[l], de, (c.recompute && l > 1) ? &w.enc[l & 1] : nullptr);
  for (int l = 0; l < c.dlayers; l++) carve_layer(cv, w.dec[l], dd, (c.recompute && l > 1) ? &w.dec[l & 1] : nullptr);
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
  const size_t gw = std::max(R * c.d, Rd * c.dd);
  w.g0 = cv.take<float>(gw);
  w.g1 = cv.take<float>(gw);
  w.dlogits = cv.take<bf16_t>(Rm * K);
  w.d_vis = cv.take<bf16_t>(R * c.dd);
  BwdNeeds nd;
  bwd_scratch_requirements(de, nd);
  bwd_scratch_requirements(dd, nd);
  auto tn = [&](size_t M, int n1, int n2) { nd.scratch = std::max(nd.scratch, gemm_tn_workspace_bytes((int)M, n1, n2)); };
  tn(R, c.d, K); tn(R, c.dd, c.d); tn(Rm, K, c.dd);
  nd.scratch = std::max(nd.scratch, layernorm_bwd_workspace_bytes((int)Rm, c.dd));
  nd.scratch = std::max(nd.scratch, colsum_workspace_bytes((int)std::max(R, Rd), std::max(std::max(c.d, c.dd), K)));
  carve_bwd_scratch(cv, w.sc, nd);
  carve_wgrad_pipe(cv, w.pipe, gw, nd);
  // row gathers of the ranged last decoder block at B > 1: residual-gradient rows [Rm, dd] and attention-output rows [Rm, H * slot] (wider than dd
  // when the heads are zero-padded to 64-wide slots)
  const size_t gath = Rm * (size_t)std::max(c.dd, dd.wa());
  w.sc.gath0 = cv.take<bf16_t>(gath);
  w.sc.gath1 = cv.take<bf16_t>(gath);
  w.total = (cv.off + 255) & ~(size_t)255;
}


static int mim_forward(const smbv_mim_cfg& c, const smbv_mim_tensors& p, const smbv_mim_io& io, void* ws, size_t ws_bytes, hipStream_t st) {
  RC(check_cfg(c));
  MimWs W;
  carve(c, ws, W);
  if (!ws || ws_bytes < W.total || ((uintptr_t)ws & 255)) { set_error("mim_forward: workspace too small or misaligned (%zu < %zu)", ws_bytes, W.total); return SMBV_ERR_WORKSPACE; }
  if (!io.pixel_values || !io.vis_idx || !io.mask_idx || !io.pos_enc || !io.pos_dec || !io.loss || !io.logits) { set_error("mim_forward: null io pointer"); return SMBV_ERR_ARG; }
  const int K = c.ts * c.ps * c.ps;
  const size_t R = (size_t)c.B * c.Nvis, Rd = (size_t)c.B * c.N, Rm = (size_t)c.B * c.Nmask;
  const BlockDims de = enc_dims(c), dd = dec_dims(c);

  // bf16 operand copies (plain and transposed) of the fp32 master weights, which change every optimizer step: one launch per 64 jobs
  SideStream prep_side;
  int prep_join_layer = -1;
  {
    PrepList pl;
    pl.cast(p.pe_w, W.pe_w, c.d, K);
    int first_block_end = 0;   // jobs [0, first_block_end): the patch-embedding weight and the first block's copies
    for (int l = 0; l < c.layers; l++) {
      RC(prep_layer(pl, to_block(p.enc[l]), W.enc[l], de, st));
      if (l == 0) first_block_end = (int)pl.jobs.size();
    }
    pl.both(p.e2d_w, W.e2d, c.d, W.e2d_t, c.dd, c.dd, c.d);
    for (int l = 0; l < c.dlayers; l++) RC(prep_layer(pl, to_block(p.dec[l]), W.dec[l], dd, st));
    pl.both(p.head_w, W.head, c.dd, W.head_t, K, K, c.dd);
    // Only what the first kernels read goes ahead of them on `st`: the patch-embedding weight and the first block's copies (17 jobs, ~30 us; round 2 put
    // the first 64 jobs there, 113 us of the critical path).  Everything else - in the order the forward reaches it - runs on the library's side stream
    // beside the brick gather, the patch-embedding product and the first block, and is joined in front of the second block (0.3 ms of HBM-bound casts
    // against ~0.5 ms of main-stream work; head-padded configurations keep everything on `st`: prep_layer's zero fills are ordered on it)
    const int njobs = (int)pl.jobs.size(), first = std::min(njobs, first_block_end);
    const bool split = njobs > first && !de.padded() && !dd.padded();
    RC(prep_weights(pl.jobs.data(), split ? first : njobs, st));
    if (split) {
      RC(prep_side.begin(st));
      RC(prep_weights(pl.jobs.data() + first, njobs - first, prep_side.st));
      prep_join_layer = 1;
      // the target statistics depend on the volume and the mask only: same side stream, long done when the loss needs them
      // (round 5 tried it elsewhere - forked beside the decoder's first blocks: +0.08 ms, profiles/r05_adamw_order_ab.txt; fused into the loss kernel, one pass over
      // the masked bricks for both: +0.08 ms, profiles/r05_stats_loss_fused_ab.txt - the head of the step is HBM-bound anyway and absorbs these 218 MB best)
      RC(brick_stats(io.pixel_values, io.mask_idx, (int)Rm, c.Nmask, W.bmean, W.brinv, c.T, c.H, c.W, c.ts, c.ps, prep_side.st));
    }
  }

  // patch embedding of the visible bricks + bias + sin-cos rows (spec :124-139, :179-192)
  RC(gather_bricks(io.pixel_values, io.vis_idx, (int)R, c.Nvis, W.a_patch, c.T, c.H, c.W, c.ts, c.ps, st));
  {
    GemmEpi e; e.bias = p.pe_b; e.rowadd = io.pos_enc; e.rowidx = io.vis_idx; e.ldr = c.d; e.out_f32 = W.x[0]; e.ld_of = c.d;
    RC(gemm_nt(W.a_patch, K, W.pe_w, K, (int)R, c.d, K, e, st));
  }
  for (int l = 0; l < c.layers; l++) {
    if (l == prep_join_layer) RC(prep_side.end(st));
    RC(layer_fwd(to_block(p.enc[l]), W.enc[l], de, W.sc, W.x[l], W.x[l + 1], l == c.layers - 1 ? W.enc_out : nullptr, 0, 0, st));
  }
  if (prep_join_layer >= c.layers) RC(prep_side.end(st));
  // encoder_to_decoder + decoder input assembly: [visible + pos_vis, mask_token + pos_mask] (spec :801-815)
  {
    GemmEpi e; e.rowadd = io.pos_dec; e.rowidx = io.vis_idx; e.ldr = c.dd; e.out_f32 = W.xd[0]; e.ld_of = c.dd; e.grp_in = c.Nvis; e.grp_out = c.N;
    RC(gemm_nt(W.enc_out, c.d, W.e2d, c.d, (int)R, c.dd, c.d, e, st));
  }
  RC(fill_mask_rows(W.xd[0], p.mask_token, io.pos_dec, io.mask_idx, c.B, c.Nmask, c.Nvis, c.dd, st));
  for (int l = 0; l < c.dlayers; l++)
    RC(layer_fwd(to_block(p.dec[l]), W.dec[l], dd, W.sc, W.xd[l], W.xd[l + 1], nullptr, l == c.dlayers - 1 ? c.Nmask : 0,
                 l == c.dlayers - 1 ? c.Nvis : 0, st));
  // last N_mask tokens -> decoder.norm -> head (spec :717-722)
  RC(layernorm_fwd(W.xd[c.dlayers], p.dnorm_w, p.dnorm_b, W.lnf, W.meanf, W.rstdf, (int)Rm, c.dd, c.dec_norm_eps, c.Nmask, c.N, c.Nvis, st));
  {
    GemmEpi e; e.bias = p.head_b; e.out_bf16 = io.logits; e.ld_ob = K;
    RC(gemm_nt(W.lnf, c.dd, W.head, c.dd, (int)Rm, K, c.dd, e, st));
  }
  // per-patch normalised targets + MSE (spec :822-897)
  if (prep_join_layer < 0) RC(brick_stats(io.pixel_values, io.mask_idx, (int)Rm, c.Nmask, W.bmean, W.brinv, c.T, c.H, c.W, c.ts, c.ps, st));
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
  WgradPipe& pipe = W.pipe;
  RC(pipe.begin());
  // (recompute: the blocks of a stack alternate between TWO activation sets; a re-run forward overwrites the set the block two launches back read, so
  // that launch is waited for in front of it - the rule layer_bwd applies to the gradient rings anyway.  Same launches, same arithmetic as the resident
  // step: bit-identical gradients)
  auto next_event = [&]() -> hipEvent_t { return events ? (hipEvent_t)events[ev++] : (ev++, (hipEvent_t) nullptr); };
  auto mark = [&]() -> int { return pipe.mark_inline(next_event(), st); };   // a group whose gradients were produced on `st`
  const int K = c.ts * c.ps * c.ps;
  const size_t R = (size_t)c.B * c.Nvis, Rd = (size_t)c.B * c.N, Rm = (size_t)c.B * c.Nmask;
  const BlockDims de = enc_dims(c), dd = dec_dims(c);

  RC(mse_bwd(io.pixel_values, io.mask_idx, (int)Rm, c.Nmask, W.bmean, W.brinv, io.logits, W.dlogits, io.grad_scale, c.T, c.H, c.W, c.ts, c.ps, st));
  // head
  {
    TnColsum cs; cs.dst[0] = gr.head_b; cs.seg = K;   // bias gradient = column sums of dlogits, from the same pass
    RC(gemm_tn(W.dlogits, K, W.lnf, c.dd, (int)Rm, K, c.dd, gr.head_w, W.sc.scratch, W.sc.scratch_bytes, st, &cs));
  }
  {
    GemmEpi e; e.out_bf16 = W.sc.d_ln; e.ld_ob = c.dd;
    RC(gemm_nt(W.dlogits, K, W.head_t, K, (int)Rm, c.dd, K, e, st));
  }
  // decoder.norm backward scatters into the last N_mask rows of every volume; visible rows get no gradient from the head
  HIPRC(hipMemsetAsync(W.g0, 0, Rd * c.dd * sizeof(float), st));
  HIPRC(hipMemsetAsync(pipe.cur(), 0, Rd * c.dd * sizeof(bf16_t), st));
  RC(layernorm_bwd(W.sc.d_ln, W.xd[c.dlayers], W.meanf, W.rstdf, p.dnorm_w, nullptr, W.g0, pipe.cur(), gr.dnorm_w, gr.dnorm_b, (int)Rm, c.dd, c.Nmask, c.N,
                   c.Nvis, W.sc.scratch, W.sc.scratch_bytes, st));
  RC(mark());
  for (int l = c.dlayers - 1; l >= 0; l--) {
    const bool last = l == c.dlayers - 1;  // its forward only produced the masked tokens (see mim_forward)
    // recompute: rebuild this block's activations from its saved input (the last block's are still in its set)
    if (c.recompute && !last) { RC(pipe.wait_for(pipe.k - 2, st)); RC(layer_fwd(to_block(p.dec[l]), W.dec[l], dd, W.sc, W.xd[l], W.xd[l + 1], nullptr, 0, 0, st)); }
    RC(layer_bwd(to_block(p.dec[l]), to_block(gr.dec[l]), W.dec[l], dd, W.sc, W.xd[l], W.g0, W.g1, pipe, last ? c.Nmask : 0, last ? c.Nvis : 0, next_event(), st));
  }
  // decoder input: mask_token gradient = sum over masked rows; visible rows flow into encoder_to_decoder
  RC(colsum_f32(W.g0, c.dd, (int)Rm, c.dd, gr.mask_token, c.Nmask, c.N, c.Nvis, W.sc.scratch, W.sc.scratch_bytes, st));
  RC(gather_rows_bf16(pipe.cur(), W.d_vis, (int)R, c.dd, c.Nvis, c.N, 0, st));
  RC(gemm_tn(W.d_vis, c.dd, W.enc_out, c.d, (int)R, c.dd, c.d, gr.e2d_w, W.sc.scratch, W.sc.scratch_bytes, st));
  {
    // (pipe.cur() was the last decoder block's output gradient, consumed by the gather above; no deferred launch reads a block's OUTPUT buffer)
    GemmEpi e; e.out_f32 = W.g1; e.ld_of = c.d; e.out_bf16 = pipe.cur(); e.ld_ob = c.d;
    RC(gemm_nt(W.d_vis, c.dd, W.e2d_t, c.dd, (int)R, c.d, c.dd, e, st));
  }
  RC(mark());
  for (int l = c.layers - 1; l >= 0; l--) {
    // recompute: the decoder ran after the encoder and does not touch the encoder's shared set, so the last encoder block's activations survive
    if (c.recompute && l != c.layers - 1) { RC(pipe.wait_for(pipe.k - 2, st)); RC(layer_fwd(to_block(p.enc[l]), W.enc[l], de, W.sc, W.x[l], W.x[l + 1], nullptr, 0, 0, st)); }
    pipe.last_block = l == 0;
    const int rcl = layer_bwd(to_block(p.enc[l]), to_block(gr.enc[l]), W.enc[l], de, W.sc, W.x[l], W.g1, W.g0, pipe, 0, 0, next_event(), st);
    pipe.last_block = false;
    RC(rcl);
  }
  // patch embedding (the volume needs no gradient)
  {
    TnColsum cs; cs.dst[0] = gr.pe_b; cs.seg = c.d;
    RC(gemm_tn(pipe.cur(), c.d, W.a_patch, K, (int)R, c.d, K, gr.pe_w, W.sc.scratch, W.sc.scratch_bytes, st, &cs));
  }
  RC(mark());
  RC(pipe.drain(st));   // the caller's stream is ordered behind every deferred launch: all gradients are final where this call's work ends
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

int smbv_mim_hidden_states(const smbv_mim_cfg* cfg, void* ws, size_t ws_bytes, float** out, int n_out) {
  if (!cfg || !out) { smbv::set_error("smbv_mim_hidden_states: null argument"); return SMBV_ERR_ARG; }
  if (int rc = smbv::check_cfg(*cfg)) return rc;
  if (n_out != cfg->layers + 1) { smbv::set_error("smbv_mim_hidden_states: n_out = %d, the encoder has %d hidden states (embeddings + one per layer)", n_out, cfg->layers + 1); return SMBV_ERR_ARG; }
  smbv::MimWs W;
  smbv::carve(*cfg, ws, W);
  if (!ws || ws_bytes < W.total || ((uintptr_t)ws & 255)) { smbv::set_error("smbv_mim_hidden_states: workspace too small or misaligned (%zu < %zu)", ws_bytes, W.total); return SMBV_ERR_WORKSPACE; }
  for (int l = 0; l <= cfg->layers; l++) out[l] = W.x[l];
  return SMBV_OK;
}

int smbv_mim_forward(const smbv_mim_cfg* cfg, const smbv_mim_tensors* params, const smbv_mim_io* io, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!cfg || !params || !io) { smbv::set_error("smbv_mim_forward: null argument"); return SMBV_ERR_ARG; }
  SMBV_CHECK_DEVICE_STATE();
  return smbv::mim_forward(*cfg, *params, *io, ws, ws_bytes, (hipStream_t)stream);
}

int smbv_mim_backward(const smbv_mim_cfg* cfg, const smbv_mim_tensors* params, const smbv_mim_tensors* grads, const smbv_mim_io* io, void* ws,
                      size_t ws_bytes, void** events, int n_events, smbv_stream stream) {
  if (!cfg || !params || !grads || !io) { smbv::set_error("smbv_mim_backward: null argument"); return SMBV_ERR_ARG; }
  SMBV_CHECK_DEVICE_STATE();
  return smbv::mim_backward(*cfg, *params, *grads, *io, ws, ws_bytes, events, n_events, (hipStream_t)stream);
}

}  // extern "C"
