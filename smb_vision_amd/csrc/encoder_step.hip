// Encoder-only forward: VideoMAEModel.forward (src/models/videomae/modeling_videomae.py:509-655) - patch
// embedding + sin-cos rows of the selected tokens, L pre-LN blocks, optional final LayerNorm (absent when config.use_mean_pooling,
// :517-520) - as used by the embedding export (src/run_inference.py:78-83: model.videomae(image), all 20,480 tokens, no mask) and
// under the classification head (spec :917-989).  cfg.train == 0 (inference): one shared set of activation buffers serves every
// layer.  cfg.train != 0 (fine-tuning, src/run_classification.py): every layer keeps its activations and smbv_encoder_backward turns the
// gradient of last_hidden_state into parameter gradients (the classification head itself is a [B, d] computation left to the caller).
#include "block.h"
#include "../../include/smbv.h"

namespace smbv {

struct EncWs {
  bf16_t* pe_w;
  LayerWs L[SMBV_MAX_LAYERS];
  bf16_t *a_patch, *lnf_bf16;
  float *x0, *x1, *meanf, *rstdf;
  float* x[SMBV_MAX_LAYERS + 1];  // train: residual stream per layer
  float *g0, *g1;                 // train: fp32 gradient ping-pong
  bf16_t* d_out_bf16;
  BwdScratch sc;
  WgradPipe pipe;                 // train: bf16 gradient ring + the rings of the deferred weight-gradient launches (block.h)
  size_t total;
};

static BlockDims edims(const smbv_encoder_cfg& c) {
  BlockDims d; d.rows = (size_t)c.B * c.Nsel; d.w = c.d; d.ff = c.ff; d.B = c.B; d.S = c.Nsel; d.H = c.heads; d.Dh = c.d / c.heads; d.eps = c.ln_eps; d.o_resid = c.train ? c.attn_o_resid : 0;
  return d;
}

static int check_cfg(const smbv_encoder_cfg& c) {
  if (c.struct_bytes != (int)sizeof(smbv_encoder_cfg)) { set_error("encoder: cfg.struct_bytes = %d, this library's smbv_encoder_cfg is %d bytes (binding built against another ABI: check smbv_version())", c.struct_bytes, (int)sizeof(smbv_encoder_cfg)); return SMBV_ERR_ARG; }
  if (c.B <= 0 || c.layers <= 0 || c.layers > SMBV_MAX_LAYERS) { set_error("encoder: bad batch/layer counts"); return SMBV_ERR_SHAPE; }
  if (c.ps <= 0 || c.ts <= 0 || c.T % c.ts || c.H % c.ps || c.W % c.ps || (c.ps & 3)) { set_error("encoder: volume %dx%dx%d not divisible by tubelet %d / patch %d", c.T, c.H, c.W, c.ts, c.ps); return SMBV_ERR_SHAPE; }
  const int N = (c.T / c.ts) * (c.H / c.ps) * (c.W / c.ps);
  if (c.N != N || c.Nsel <= 0 || c.Nsel > N) { set_error("encoder: token counts N=%d Nsel=%d inconsistent with the volume (%d)", c.N, c.Nsel, N); return SMBV_ERR_SHAPE; }
  if (c.heads <= 0 || c.d % c.heads || c.d / c.heads > 64 || ((c.d / c.heads) & 7)) { set_error("encoder: head_dim must be a multiple of 8 and <= 64 (d=%d/%d)", c.d, c.heads); return SMBV_ERR_SHAPE; }
  if ((c.d & 7) || (c.ff & 7) || ((c.ts * c.ps * c.ps) & 7)) { set_error("encoder: widths must be multiples of 8"); return SMBV_ERR_SHAPE; }
  return SMBV_OK;
}

static void carve(const smbv_encoder_cfg& c, void* base, EncWs& w) {
  Carver cv{(char*)base};
  const int K = c.ts * c.ps * c.ps;
  const size_t R = (size_t)c.B * c.Nsel;
  const BlockDims de = edims(c);
  w.pe_w = cv.take<bf16_t>((size_t)c.d * K);
  if (c.train) {
    for (int l = 0; l < c.layers; l++) carve_layer(cv, w.L[l], de);
    w.a_patch = cv.take<bf16_t>(R * K);
    for (int l = 0; l <= c.layers; l++) w.x[l] = cv.take<float>(R * c.d);
    w.x0 = w.x[0]; w.x1 = nullptr;
    w.lnf_bf16 = cv.take<bf16_t>(R * c.d);
    w.meanf = cv.take<float>(R);
    w.rstdf = cv.take<float>(R);
    w.g0 = cv.take<float>(R * c.d);
    w.g1 = cv.take<float>(R * c.d);
    w.d_out_bf16 = cv.take<bf16_t>(R * c.d);
    BwdNeeds nd;
    bwd_scratch_requirements(de, nd);
    nd.scratch = std::max(nd.scratch, gemm_tn_workspace_bytes((int)R, c.d, K));
    nd.scratch = std::max(nd.scratch, colsum_workspace_bytes((int)R, std::max(c.d, K)));
    w.sc = BwdScratch{};
    carve_bwd_scratch(cv, w.sc, nd);
    carve_wgrad_pipe(cv, w.pipe, R * c.d, nd);
    w.sc.gath0 = cv.take<bf16_t>(16);
    w.sc.gath1 = cv.take<bf16_t>(16);
    w.total = (cv.off + 255) & ~(size_t)255;
    return;
  }
  carve_layer(cv, w.L[0], de);  // weights of layer 0 + the ONE set of activation buffers
  const size_t ww = (size_t)de.wa() * c.d, fw = (size_t)c.ff * c.d;   // (wa = heads x slot: wider than d when the heads are zero-padded to 64-wide slots)
  for (int l = 1; l < c.layers; l++) {
    LayerWs& L = w.L[l];
    L = w.L[0];                                    // share every activation buffer ...
    L.wqkv = cv.take<bf16_t>(3 * ww);             // ... but not the bf16 operand copies of the weights
    L.wqkv_t = w.L[0].wqkv_t;                     // transposed copies exist for dgrads only: all layers dump theirs into layer 0's
    L.proj = cv.take<bf16_t>(ww);
    L.proj_t = w.L[0].proj_t;
    L.fc1 = cv.take<bf16_t>(fw);
    L.fc1_t = w.L[0].fc1_t;
    L.fc2 = cv.take<bf16_t>(fw);
    L.fc2_t = w.L[0].fc2_t;
    L.bqkv = cv.take<float>(3 * (size_t)de.wa());
  }
  w.a_patch = cv.take<bf16_t>(R * K);
  w.x0 = cv.take<float>(R * c.d);
  w.x1 = cv.take<float>(R * c.d);
  w.lnf_bf16 = cv.take<bf16_t>(R * c.d);
  w.meanf = cv.take<float>(R);
  w.rstdf = cv.take<float>(R);
  w.sc = BwdScratch{};
  w.sc.gath0 = cv.take<bf16_t>(16);
  w.total = (cv.off + 255) & ~(size_t)255;
}

static int encoder_forward(const smbv_encoder_cfg& c, const smbv_encoder_tensors& p, const smbv_encoder_io& io, void* ws, size_t ws_bytes, hipStream_t st) {
  RC(check_cfg(c));
  EncWs W;
  carve(c, ws, W);
  if (!ws || ws_bytes < W.total || ((uintptr_t)ws & 255)) { set_error("encoder_forward: workspace too small or misaligned (%zu < %zu)", ws_bytes, W.total); return SMBV_ERR_WORKSPACE; }
  if (!io.pixel_values || !io.sel_idx || !io.pos_enc || !io.last_hidden_state) { set_error("encoder_forward: null io pointer"); return SMBV_ERR_ARG; }
  if (c.final_ln && (!p.lnf_w || !p.lnf_b)) { set_error("encoder_forward: final_ln set but no layernorm parameters"); return SMBV_ERR_ARG; }
  const int K = c.ts * c.ps * c.ps;
  const size_t R = (size_t)c.B * c.Nsel;
  const BlockDims de = edims(c);
  auto blk = [&](int l) {
    const smbv_layer_ptrs& q = p.enc[l];
    BlockPtrs b;
    b.ln1_w = q.ln1_w; b.ln1_b = q.ln1_b; b.q_w = q.q_w; b.k_w = q.k_w; b.v_w = q.v_w; b.q_b = q.q_b; b.k_b = nullptr; b.v_b = q.v_b;
    b.proj_w = q.proj_w; b.proj_b = q.proj_b; b.ln2_w = q.ln2_w; b.ln2_b = q.ln2_b; b.fc1_w = q.fc1_w; b.fc1_b = q.fc1_b; b.fc2_w = q.fc2_w; b.fc2_b = q.fc2_b;
    return b;
  };
  {
    PrepList pl;
    pl.cast(p.pe_w, W.pe_w, c.d, K);
    for (int l = 0; l < c.layers; l++) RC(prep_layer(pl, blk(l), W.L[l], de, st));
    RC(prep_weights(pl.jobs.data(), (int)pl.jobs.size(), st));
  }
  RC(gather_bricks(io.pixel_values, io.sel_idx, (int)R, c.Nsel, W.a_patch, c.T, c.H, c.W, c.ts, c.ps, st));
  float* xin = W.x0;
  float* xout = W.x1;
  {
    GemmEpi e; e.bias = p.pe_b; e.rowadd = io.pos_enc; e.rowidx = io.sel_idx; e.ldr = c.d; e.out_f32 = xin; e.ld_of = c.d;
    RC(gemm_nt(W.a_patch, K, W.pe_w, K, (int)R, c.d, K, e, st));
  }
  if (c.train) {
    for (int l = 0; l < c.layers; l++) RC(layer_fwd(blk(l), W.L[l], de, W.sc, W.x[l], W.x[l + 1], nullptr, 0, 0, st));
    xin = W.x[c.layers];
    if (!c.final_ln) HIPRC(hipMemcpyAsync(io.last_hidden_state, xin, R * c.d * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    for (int l = 0; l < c.layers; l++) {
      const bool last = l == c.layers - 1;
      float* dst = (last && !c.final_ln) ? io.last_hidden_state : xout;
      RC(layer_fwd(blk(l), W.L[l], de, W.sc, xin, dst, nullptr, 0, 0, st));
      xout = xin;
      xin = dst;
    }
  }
  if (c.final_ln)
    RC(layernorm_fwd(xin, p.lnf_w, p.lnf_b, W.lnf_bf16, W.meanf, W.rstdf, (int)R, c.d, c.final_ln_eps, 0, 0, 0, st, io.last_hidden_state));
  return SMBV_OK;
}

// Gradients of every encoder parameter from d(last_hidden_state) (fp32 [B*Nsel, d]); needs the workspace of a train-mode forward.
static int encoder_backward(const smbv_encoder_cfg& c, const smbv_encoder_tensors& p, const smbv_encoder_tensors& gr, const smbv_encoder_io& io,
                            const float* d_out, void* ws, size_t ws_bytes, hipStream_t st) {
  RC(check_cfg(c));
  if (!c.train) { set_error("encoder_backward: the forward ran in inference mode (cfg.train == 0)"); return SMBV_ERR_ARG; }
  EncWs W;
  carve(c, ws, W);
  if (!ws || ws_bytes < W.total || ((uintptr_t)ws & 255)) { set_error("encoder_backward: workspace too small or misaligned"); return SMBV_ERR_WORKSPACE; }
  if (!d_out) { set_error("encoder_backward: null gradient"); return SMBV_ERR_ARG; }
  const int K = c.ts * c.ps * c.ps;
  const size_t R = (size_t)c.B * c.Nsel;
  const BlockDims de = edims(c);
  auto blk = [&](const smbv_encoder_tensors& t, int l) {
    const smbv_layer_ptrs& q = t.enc[l];
    BlockPtrs b;
    b.ln1_w = q.ln1_w; b.ln1_b = q.ln1_b; b.q_w = q.q_w; b.k_w = q.k_w; b.v_w = q.v_w; b.q_b = q.q_b; b.k_b = nullptr; b.v_b = q.v_b;
    b.proj_w = q.proj_w; b.proj_b = q.proj_b; b.ln2_w = q.ln2_w; b.ln2_b = q.ln2_b; b.fc1_w = q.fc1_w; b.fc1_b = q.fc1_b; b.fc2_w = q.fc2_w; b.fc2_b = q.fc2_b;
    return b;
  };
  WgradPipe& pipe = W.pipe;
  RC(pipe.begin());
  if (c.final_ln) {
    RC(cast_f32_bf16(d_out, W.d_out_bf16, R * c.d, st));
    RC(layernorm_bwd(W.d_out_bf16, W.x[c.layers], W.meanf, W.rstdf, p.lnf_w, nullptr, W.g0, pipe.cur(), gr.lnf_w, gr.lnf_b, (int)R, c.d, 0, 0, 0, W.sc.scratch,
                     W.sc.scratch_bytes, st));
  } else {
    HIPRC(hipMemcpyAsync(W.g0, d_out, R * c.d * sizeof(float), hipMemcpyDeviceToDevice, st));
    RC(cast_f32_bf16(d_out, pipe.cur(), R * c.d, st));
  }
  for (int l = c.layers - 1; l >= 0; l--)
    RC(layer_bwd(blk(p, l), blk(gr, l), W.L[l], de, W.sc, W.x[l], W.g0, W.g1, pipe, 0, 0, nullptr, st));   // fp32 gradient in / out: W.g0; bf16: the pipe's ring
  {
    TnColsum cs; cs.dst[0] = gr.pe_b; cs.seg = c.d;
    RC(gemm_tn(pipe.cur(), c.d, W.a_patch, K, (int)R, c.d, K, gr.pe_w, W.sc.scratch, W.sc.scratch_bytes, st, &cs));
  }
  RC(pipe.drain(st));
  (void)io;
  return SMBV_OK;
}

}  // namespace smbv

extern "C" {

size_t smbv_encoder_workspace_bytes(const smbv_encoder_cfg* cfg) {
  if (!cfg || smbv::check_cfg(*cfg)) return 0;
  smbv::EncWs W;
  smbv::carve(*cfg, nullptr, W);
  return W.total;
}

int smbv_encoder_hidden_states(const smbv_encoder_cfg* cfg, void* ws, size_t ws_bytes, float** out, int n_out) {
  if (!cfg || !out) { smbv::set_error("smbv_encoder_hidden_states: null argument"); return SMBV_ERR_ARG; }
  if (int rc = smbv::check_cfg(*cfg)) return rc;
  if (!cfg->train) { smbv::set_error("smbv_encoder_hidden_states: the inference forward (cfg.train == 0) shares one activation set between its layers; run the forward with cfg.train = 1"); return SMBV_ERR_ARG; }
  if (n_out != cfg->layers + 1) { smbv::set_error("smbv_encoder_hidden_states: n_out = %d, the encoder has %d hidden states (embeddings + one per layer)", n_out, cfg->layers + 1); return SMBV_ERR_ARG; }
  smbv::EncWs W;
  smbv::carve(*cfg, ws, W);
  if (!ws || ws_bytes < W.total || ((uintptr_t)ws & 255)) { smbv::set_error("smbv_encoder_hidden_states: workspace too small or misaligned (%zu < %zu)", ws_bytes, W.total); return SMBV_ERR_WORKSPACE; }
  for (int l = 0; l <= cfg->layers; l++) out[l] = W.x[l];
  return SMBV_OK;
}

int smbv_encoder_forward(const smbv_encoder_cfg* cfg, const smbv_encoder_tensors* params, const smbv_encoder_io* io, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!cfg || !params || !io) { smbv::set_error("smbv_encoder_forward: null argument"); return SMBV_ERR_ARG; }
  SMBV_CHECK_DEVICE_STATE();
  return smbv::encoder_forward(*cfg, *params, *io, ws, ws_bytes, (hipStream_t)stream);
}

int smbv_encoder_backward(const smbv_encoder_cfg* cfg, const smbv_encoder_tensors* params, const smbv_encoder_tensors* grads, const smbv_encoder_io* io,
                          const float* d_last_hidden_state, void* ws, size_t ws_bytes, smbv_stream stream) {
  if (!cfg || !params || !grads || !io) { smbv::set_error("smbv_encoder_backward: null argument"); return SMBV_ERR_ARG; }
  SMBV_CHECK_DEVICE_STATE();
  return smbv::encoder_backward(*cfg, *params, *grads, *io, d_last_hidden_state, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
