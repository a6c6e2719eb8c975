// Transformer block forward / backward sequencing (see block.h).
#include "block.h"

namespace smbv {
extern int g_variant;
extern int g_ab;

void carve_layer(Carver& c, LayerWs& L, const BlockDims& d, const LayerWs* share_acts) {
  const size_t w = d.w, wa = d.wa(), ff = d.ff, rows = d.rows;
  if (share_acts) L = *share_acts;  // recompute mode: ONE set of activation buffers for all blocks of a stack; the weight copies stay per block
  L.wqkv = c.take<bf16_t>(3 * wa * w);
  L.wqkv_t = c.take<bf16_t>(3 * wa * w);
  L.proj = c.take<bf16_t>(w * wa);
  L.proj_t = c.take<bf16_t>(w * wa);
  L.fc1 = c.take<bf16_t>(ff * w);
  L.fc1_t = c.take<bf16_t>(ff * w);
  L.fc2 = c.take<bf16_t>(ff * w);
  L.fc2_t = c.take<bf16_t>(ff * w);
  L.bqkv = c.take<float>(3 * wa);
  if (share_acts) return;
  L.ln1 = c.take<bf16_t>(rows * w);
  L.qkv = c.take<bf16_t>(rows * 3 * wa);
  L.attn_o = c.take<bf16_t>(rows * wa);
  L.attn_o_lo = (d.o_resid && !d.attn_fp8) ? c.take<bf16_t>(rows * wa) : nullptr;
  L.ln2 = c.take<bf16_t>(rows * w);
  L.pre = c.take<bf16_t>(rows * ff);
  L.act = c.take<bf16_t>(rows * ff);
  L.mean1 = c.take<float>(rows);
  L.rstd1 = c.take<float>(rows);
  L.mean2 = c.take<float>(rows);
  L.rstd2 = c.take<float>(rows);
  L.lse = c.take<float>((size_t)d.B * d.H * d.S);
  L.x_mid = c.take<float>(rows * w);
}

int prep_layer(PrepList& pl, const BlockPtrs& p, LayerWs& L, const BlockDims& d, hipStream_t st) {
  const int w = d.w, ff = d.ff, wa = d.wa(), Dh = d.Dh, hs = d.slot();
  const size_t ww = (size_t)w * w;
  if (!d.padded()) {
    pl.both(p.q_w, L.wqkv, w, L.wqkv_t, 3 * w, w, w);
    pl.both(p.k_w, L.wqkv + ww, w, L.wqkv_t + w, 3 * w, w, w);
    pl.both(p.v_w, L.wqkv + 2 * ww, w, L.wqkv_t + 2 * w, 3 * w, w, w);
    pl.both(p.proj_w, L.proj, w, L.proj_t, w, w, w);
    pl.add(p.q_b, L.bqkv, 1, w, w, w, 2);
    if (p.k_b) pl.add(p.k_b, L.bqkv + w, 1, w, w, w, 2);
    else pl.add(nullptr, L.bqkv + w, 1, w, w, w, 3);  // VideoMAE: key has no bias (spec :261)
    pl.add(p.v_b, L.bqkv + 2 * w, 1, w, w, w, 2);
  } else {
    // heads of Dh < 64 columns live in 64-column slots; everything outside the real columns is zero
    HIPRC(hipMemsetAsync(L.wqkv, 0, (size_t)3 * wa * w * sizeof(bf16_t), st));
    HIPRC(hipMemsetAsync(L.wqkv_t, 0, (size_t)3 * wa * w * sizeof(bf16_t), st));
    HIPRC(hipMemsetAsync(L.proj, 0, (size_t)w * wa * sizeof(bf16_t), st));
    HIPRC(hipMemsetAsync(L.proj_t, 0, (size_t)w * wa * sizeof(bf16_t), st));
    HIPRC(hipMemsetAsync(L.bqkv, 0, (size_t)3 * wa * sizeof(float), st));
    const float* wsrc[3] = {p.q_w, p.k_w, p.v_w};
    const float* bsrc[3] = {p.q_b, p.k_b, p.v_b};
    for (int sec = 0; sec < 3; sec++)
      for (int h = 0; h < d.H; h++) {
        const float* src = wsrc[sec] + (size_t)h * Dh * w;                       // rows h*Dh .. of [w_out, w_in]
        pl.cast(src, L.wqkv + ((size_t)sec * wa + h * hs) * w, Dh, w);
        pl.tcast(src, L.wqkv_t + (size_t)sec * wa + h * hs, Dh, w, 3 * wa);
        if (bsrc[sec]) pl.add(bsrc[sec] + h * Dh, L.bqkv + sec * wa + h * hs, 1, Dh, Dh, Dh, 2);
      }
    for (int h = 0; h < d.H; h++) {
      // proj weight [w, w_in = H*Dh]: column block h*Dh.. -> padded column block h*64..
      pl.add(p.proj_w + h * Dh, L.proj + h * hs, w, Dh, w, wa, 0);
      pl.add(p.proj_w + h * Dh, L.proj_t + (size_t)h * hs * w, w, Dh, w, w, 1);
    }
  }
  pl.both(p.fc1_w, L.fc1, w, L.fc1_t, ff, ff, w);
  pl.both(p.fc2_w, L.fc2, ff, L.fc2_t, w, w, ff);
  return SMBV_OK;
}

void bwd_scratch_requirements(const BlockDims& d, BwdNeeds& n) {
  const size_t rows = d.rows, w = d.w, wa = d.wa(), ff = d.ff;
  n.act = std::max(n.act, rows * ff);
  n.we = std::max(n.we, rows * std::max(w, wa));
  n.qkv = std::max(n.qkv, rows * 3 * wa);
  if (d.padded()) { n.gpad = std::max(n.gpad, 3 * wa * w); n.gpad_b = std::max(n.gpad_b, 3 * wa); }
  n.delta = std::max(n.delta, (size_t)d.B * d.H * d.S);
  auto tn = [&](size_t M, size_t n1, size_t n2) { n.wscratch = std::max(n.wscratch, gemm_tn_workspace_bytes((int)M, (int)n1, (int)n2)); };
  tn(rows, w, ff); tn(rows, ff, w); tn(rows, w, wa); tn(rows, 3 * wa, w); tn(rows, w, w);
  if (!(w % 256) && !(ff % 256) && !(wa % 256)) {   // the last block's grouped launch as two token halves per tile (gemm_tn_grouped): the first half's tiles pass through here
    const size_t tiles = (2 * w * ff + 3 * wa * w + wa * w) / (256 * 256);
    if (tiles <= 128) n.wscratch = std::max(n.wscratch, gemm_tn_grouped_split_bytes((int)tiles));
  }
  n.scratch = std::max(n.scratch, layernorm_bwd_workspace_bytes((int)rows, (int)w));
  n.attn_split = std::max(n.attn_split, attention_bwd_split_workspace_max(d.B, d.S, d.H, d.slot()));
  if (d.attn_fp8) n.f8 = std::max(n.f8, attention_fp8_workspace_bytes(d.B, d.S, d.H));
  if (d.attn_fp8 & 2) n.f8 = std::max(n.f8, attention_bwd_fp8_workspace_bytes(d.B, d.S, d.H));
}

void carve_bwd_scratch(Carver& cv, BwdScratch& sc, const BwdNeeds& n) {
  sc.d_ln = cv.take<bf16_t>(n.we);
  sc.d_o = cv.take<bf16_t>(n.we);
  sc.delta = cv.take<float>(n.delta);
  sc.gpad = n.gpad ? cv.take<float>(n.gpad) : nullptr;
  sc.gpad_b = n.gpad_b ? cv.take<float>(n.gpad_b) : nullptr;
  sc.scratch = cv.take<char>(n.scratch);
  sc.scratch_bytes = n.scratch;
  sc.wscratch = cv.take<char>(n.wscratch);
  sc.wscratch_bytes = n.wscratch;
  sc.f8ws = n.f8 ? cv.take<char>(n.f8) : nullptr;
  sc.f8ws_bytes = n.f8;
  sc.attn_split = n.attn_split ? cv.take<char>(n.attn_split) : nullptr;
  sc.attn_split_bytes = n.attn_split;
}

int layer_fwd(const BlockPtrs& p, LayerWs& L, const BlockDims& d, BwdScratch& sc, const float* x_in, float* x_out, bf16_t* x_out_bf16, int Sq, int qoff,
              hipStream_t st) {
  const int w = d.w, ff = d.ff, wa = d.wa(), B = d.B, S = d.S, H = d.H;
  const int M = (int)d.rows;
  RC(layernorm_fwd(x_in, p.ln1_w, p.ln1_b, L.ln1, L.mean1, L.rstd1, M, w, d.eps, 0, 0, 0, st));
  {
    GemmEpi e; e.bias = L.bqkv; e.out_bf16 = L.qkv; e.ld_ob = 3 * wa;
    // V-JEPA: the rotary embedding of q and k rides in the product's epilogue (gemm_nt: the ring kernel's EPI 3, else a rope3d launch behind the product)
    if (d.rope_grid > 0) { e.rope_table = d.rope_table; e.rope_pos = d.pos; e.rope_S = S; e.rope_grid = d.rope_grid; e.rope_Dh = d.Dh; e.rope_hs = d.slot(); e.rope_wa = wa; e.rope_rows = d.rope_rows; }
    RC(gemm_nt(L.ln1, w, L.wqkv, w, M, 3 * wa, w, e, st));
  }
  if (d.attn_fp8 && d.slot() != 64) { set_error("layer_fwd: the fp8 attention forward is built for 64-wide head slots (head_dim %d)", d.Dh); return SMBV_ERR_SHAPE; }
  if (d.attn_fp8) RC(attention_fwd_fp8(L.qkv, L.qkv + wa, L.qkv + 2 * wa, 3 * wa, L.attn_o, wa, L.lse, B, S, H, 64, d.scale(), Sq, qoff, sc.f8ws, sc.f8ws_bytes, st));
  else RC(attention_fwd(L.qkv, L.qkv + wa, L.qkv + 2 * wa, 3 * wa, L.attn_o, wa, L.lse, B, S, H, d.slot(), d.scale(), Sq, qoff, st, L.attn_o_lo));
  if (Sq > 0) {
    // ranged tail: projection + MLP on the Sq tokens only; x_mid / x_out keep the full [B*S, w] layout, ln2 / pre / act are compact
    const int Mq = B * Sq;
    const bf16_t* a_rows = L.attn_o + (size_t)qoff * wa;  // contiguous for B == 1
    if (B > 1) { RC(gather_rows_bf16(L.attn_o, sc.gath0, Mq, wa, Sq, S, qoff, st)); a_rows = sc.gath0; }
    {
      GemmEpi e; e.bias = p.proj_b; e.rowadd = x_in; e.ldr = w; e.radd_grp_in = Sq; e.radd_grp_out = S; e.radd_off = qoff;
      e.out_f32 = L.x_mid + (size_t)qoff * w; e.ld_of = w; e.grp_in = Sq; e.grp_out = S;
      RC(gemm_nt(a_rows, wa, L.proj, wa, Mq, w, wa, e, st));
    }
    RC(layernorm_fwd(L.x_mid, p.ln2_w, p.ln2_b, L.ln2, L.mean2, L.rstd2, Mq, w, d.eps, Sq, S, qoff, st));
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
  {
    GemmEpi e; e.bias = p.proj_b; e.rowadd = x_in; e.ldr = w; e.out_f32 = L.x_mid; e.ld_of = w;
    RC(gemm_nt(L.attn_o, wa, L.proj, wa, M, w, wa, e, st));
  }
  RC(layernorm_fwd(L.x_mid, p.ln2_w, p.ln2_b, L.ln2, L.mean2, L.rstd2, M, w, d.eps, 0, 0, 0, st));
  {
    GemmEpi e; e.bias = p.fc1_b; e.act_gelu = 1; e.aux_pre = L.pre; e.ld_aux = ff; e.out_bf16 = L.act; e.ld_ob = ff;
    RC(gemm_nt(L.ln2, w, L.fc1, w, M, ff, w, e, st));
  }
  {
    GemmEpi e; e.bias = p.fc2_b; e.rowadd = L.x_mid; e.ldr = w; e.out_f32 = x_out; e.ld_of = w; e.out_bf16 = x_out_bf16; e.ld_ob = w;
    RC(gemm_nt(L.act, ff, L.fc2, ff, M, w, ff, e, st));
  }
  return SMBV_OK;
}

void carve_wgrad_pipe(Carver& cv, WgradPipe& pipe, size_t gb_elems, const BwdNeeds& n) {
  for (int i = 0; i < 6; i++) pipe.gb[i] = cv.take<bf16_t>(gb_elems);
  for (int i = 0; i < 2; i++) {
    pipe.d_act[i] = cv.take<bf16_t>(n.act);
    pipe.d_qkv[i] = cv.take<bf16_t>(n.qkv);
    pipe.ln1_part[i] = cv.take<char>(n.scratch);
    pipe.ln2_part[i] = cv.take<char>(n.scratch);
  }
  pipe.part_bytes = n.scratch;
}

int WgradPipe::begin() {
  fj = (g_variant & 16384) ? nullptr : fork_join_for_current_device();
  k = 0;
  waited = -1;
  for (bool& x : deferred) x = false;
  return SMBV_OK;
}
int WgradPipe::wait_for(int launch, hipStream_t st) {
  if (!fj || launch < 0 || launch <= waited || launch < k - 4 || !deferred[launch & 3]) return SMBV_OK;
  RC(stream_wait_locked(fj, st, fj->wdone[launch & 3]));
  waited = launch;
  return SMBV_OK;
}
int WgradPipe::drain(hipStream_t st) {
  if (k > 0) RC(wait_for(k - 1, st));   // the side stream is in order: its last launch finishing means all of them have
  return SMBV_OK;
}
int WgradPipe::mark_inline(hipEvent_t ev, hipStream_t st) {
  if (!ev) return SMBV_OK;
  const bool pending = fj && k > 0 && deferred[(k - 1) & 3] && waited < k - 1;
  if (!pending) { HIPRC(hipEventRecord(ev, st)); return SMBV_OK; }
  RC(stream_edge(fj, fj->wfork, st, fj->wside));
  HIPRC(hipEventRecord(ev, fj->wside));
  return SMBV_OK;
}

namespace {
struct BwdBufs {
  float *g, *g2;                       // fp32 residual gradient: in / out, mid
  bf16_t *gb_in, *gb_mid, *gb_out;     // its bf16 copies (gb_out may alias gb_in when nothing deferred reads gb_in)
  bf16_t *d_act, *d_qkv;
  void *ln1_part, *ln2_part;
  size_t part_bytes;
};

// defer != null: the weight-gradient products are collected and issued as one grouped launch on defer's side stream at the end (see block.h)
int layer_bwd_core(const BlockPtrs& p, const BlockPtrs& gr, LayerWs& L, const BlockDims& d, BwdScratch& sc, const float* x_in, const BwdBufs& bf, int Sq, int qoff,
                   hipStream_t st, WgradPipe* defer, int launch_no, hipEvent_t dp_event, WgradPipe* pipe) {
  const int w = d.w, ff = d.ff, wa = d.wa(), B = d.B, S = d.S, H = d.H;
  const int M = (int)d.rows;
  const bool ranged = Sq > 0;
  const int Mq = ranged ? B * Sq : M;
  float *g = bf.g, *g2 = bf.g2;
  bf16_t *gb = bf.gb_in, *gb2 = bf.gb_mid;
  SideStream side;
  TnJob jobs[6];   // fc2, fc1, proj, fused q|k|v - or q, k, v separately when their gradients are not adjacent in the arena (V-JEPA: biases in between)
  int nj = 0;
  // one weight (+ bias: the column sums ride along, gemm.hip) gradient product: collected for the grouped launch, or issued on the side stream beside
  // the input-gradient product that shares its dY
  auto wgrad = [&](const bf16_t* X, int ldx, const bf16_t* Y, int ldy, int Mr, int N1, int N2, float* out, const TnColsum& cs) -> int {
    if (defer) {
      TnJob& j = jobs[nj++];
      j.X = X; j.Y = Y; j.out = out; j.ldx = ldx; j.ldy = ldy; j.M = Mr; j.N1 = N1; j.N2 = N2; j.seg = cs.seg; j.tile0 = 0;
      for (int i = 0; i < 3; i++) j.sum_dst[i] = cs.dst[i];
      return SMBV_OK;
    }
    return gemm_tn(X, ldx, Y, ldy, Mr, N1, N2, out, sc.wscratch, sc.wscratch_bytes, side.st, &cs);
  };
  // ---- MLP (on the ranged rows only when Sq > 0: ln2 / pre / act are compact then) ----
  const bf16_t* gbc = gb;
  if (ranged) {
    gbc = gb + (size_t)qoff * w;  // contiguous for B == 1
    if (B > 1) { RC(gather_rows_bf16(gb, sc.gath0, Mq, w, Sq, S, qoff, st)); gbc = sc.gath0; }
  }
  if (!defer) RC(side.begin(st));
  {
    TnColsum cs; cs.dst[0] = gr.fc2_b; cs.seg = w;
    RC(wgrad(gbc, w, L.act, ff, Mq, w, ff, gr.fc2_w, cs));
  }
  {
    GemmEpi e; e.dgelu_pre = L.pre; e.ld_pre = ff; e.out_bf16 = bf.d_act; e.ld_ob = ff;
    RC(gemm_nt(gbc, w, L.fc2_t, w, Mq, ff, w, e, st));
  }
  if (!defer) RC(side.begin(st));  // d_act is ready
  {
    TnColsum cs; cs.dst[0] = gr.fc1_b; cs.seg = ff;
    RC(wgrad(bf.d_act, ff, L.ln2, w, Mq, ff, w, gr.fc1_w, cs));
  }
  {
    GemmEpi e; e.out_bf16 = sc.d_ln; e.ld_ob = w;
    RC(gemm_nt(bf.d_act, ff, L.fc1_t, ff, Mq, w, ff, e, st));
  }
  // ---- LN2 (+ residual): g2 = g + LNbwd(d_ln); every non-ranged row of g2 is zero ----
  if (ranged) {
    HIPRC(hipMemsetAsync(g2, 0, (size_t)M * w * sizeof(float), st));
    HIPRC(hipMemsetAsync(gb2, 0, (size_t)M * w * sizeof(bf16_t), st));
    RC(layernorm_bwd(sc.d_ln, L.x_mid, L.mean2, L.rstd2, p.ln2_w, g, g2, gb2, gr.ln2_w, gr.ln2_b, Mq, w, Sq, S, qoff, bf.ln2_part, bf.part_bytes, st, false));
  } else {
    RC(layernorm_bwd(sc.d_ln, L.x_mid, L.mean2, L.rstd2, p.ln2_w, g, g2, gb2, gr.ln2_w, gr.ln2_b, M, w, 0, 0, 0, bf.ln2_part, bf.part_bytes, st, false));
  }
  // (the partials' reduction into the LN2 weight / bias gradients follows on the side stream)
  // ---- attention output projection ----
  const bf16_t* g2c = gb2;
  const bf16_t* aoc = L.attn_o;
  if (ranged) {
    g2c = gb2 + (size_t)qoff * w;
    aoc = L.attn_o + (size_t)qoff * wa;
    if (B > 1) {
      RC(side.end(st));  // the MLP weight-gradient products read the gathered rows these two copies overwrite
      RC(gather_rows_bf16(gb2, sc.gath0, Mq, w, Sq, S, qoff, st));
      RC(gather_rows_bf16(L.attn_o, sc.gath1, Mq, wa, Sq, S, qoff, st));
      g2c = sc.gath0; aoc = sc.gath1;
    }
  }
  if (!defer) {
    RC(side.begin(st));  // gb2 (and, ranged, its gathered copy) is ready
    RC(layernorm_bwd_reduce(bf.ln2_part, Mq, w, gr.ln2_w, gr.ln2_b, side.st));   // (-0.04 ms per MIM step against running it on `st` behind the LN2 backward)
  }
  {
    TnColsum cs; cs.dst[0] = gr.proj_b; cs.seg = w;
    if (!d.padded()) {
      RC(wgrad(g2c, w, aoc, wa, Mq, w, wa, gr.proj_w, cs));
    } else {
      RC(gemm_tn(g2c, w, aoc, wa, Mq, w, wa, sc.gpad, sc.wscratch, sc.wscratch_bytes, side.st, &cs));   // [w, H*64] -> [w, H*Dh]
      RC(map_copy_f32(gr.proj_w, sc.gpad, (size_t)w * w, d.Dh, 64, side.st));
    }
  }
  bool delta_fused = false;
  {
    GemmEpi e; e.out_bf16 = sc.d_o; e.ld_ob = wa;
    if (ranged) { e.out_bf16 = sc.d_o + (size_t)qoff * wa; e.grp_in = Sq; e.grp_out = S; }
    // delta = rowsum(dO * O) per (token, head) comes out of this product's epilogue (gemm.hip) instead of a launch of its own in attention_bwd
    e.dot_with = L.attn_o + (ranged ? (size_t)qoff * wa : 0); e.ld_dot = wa; e.dot_dh = d.slot(); e.dot_rows = ranged ? Sq : S; e.dot_out = sc.delta;
    if (L.attn_o_lo) e.dot_with_lo = L.attn_o_lo + (ranged ? (size_t)qoff * wa : 0);
    if (!(g_variant & 4096) && gemm_nt_rowdot_ok(w, w, Mq, wa, w, e)) delta_fused = true;
    else { e.dot_with = nullptr; e.dot_with_lo = nullptr; e.dot_out = nullptr; }
    RC(gemm_nt(g2c, w, L.proj_t, w, Mq, wa, w, e, st));
  }
  // ---- attention (dQ exists only for the ranged queries; dK/dV for every token) ----
  RopeBwd rope;
  rope.table = d.rope_table; rope.pos = d.pos; rope.grid = d.rope_grid; rope.Dh = d.Dh; rope.rows = d.rope_rows;
  if (ranged) HIPRC(hipMemsetAsync(bf.d_qkv, 0, (size_t)M * 3 * wa * sizeof(bf16_t), st));
  RC(attention_bwd(L.qkv, L.qkv + wa, L.qkv + 2 * wa, 3 * wa, L.attn_o, sc.d_o, wa, L.lse, sc.delta, bf.d_qkv, bf.d_qkv + wa, bf.d_qkv + 2 * wa, 3 * wa, B,
                   S, H, d.slot(), d.scale(), delta_fused ? 6 : 7, Sq, qoff, st, L.attn_o_lo, (d.attn_fp8 & 2) && d.slot() == 64 ? sc.f8ws : nullptr, sc.f8ws_bytes, sc.attn_split, sc.attn_split_bytes,
                   d.rope_grid > 0 ? &rope : nullptr));   // V-JEPA: dQ / dK leave through the transposed rotary map in the kernel's stores
  // ---- QKV ----
  if (!defer) {
    // (round 3 joined the three products forked above here.  Since round 4 every buffer they read - gb_in, gb_mid, d_act, d_qkv, the saved activations -
    // is a ring entry or an activation that nothing on `st` rewrites before the block's final join: LN1 backward writes g, gb_OUT and its own partials, the
    // QKV input-gradient product writes sc.d_ln, which no side product reads.  Without the join the block's tail no longer waits for a weight-gradient
    // product that the attention backward starved: what makes the inline form the right one for the LAST block of a backward, mim_step.hip.)
    if (g_ab & 1) RC(side.end(st));   // A/B: the round-3 join
    RC(side.begin(st));  // d_qkv is ready
  }
  const size_t ww = (size_t)w * w;
  const bool contiguous = gr.k_w == gr.q_w + ww && gr.v_w == gr.k_w + ww;
  if (!d.padded()) {
    if (contiguous) {
      TnColsum cs; cs.dst[0] = gr.q_b; cs.dst[1] = gr.k_b; cs.dst[2] = gr.v_b; cs.seg = w;   // VideoMAE has no key bias: null drops that segment
      RC(wgrad(bf.d_qkv, 3 * w, L.ln1, w, M, 3 * w, w, gr.q_w, cs));
    } else {
      float* wdst[3] = {gr.q_w, gr.k_w, gr.v_w};
      float* bdst[3] = {gr.q_b, gr.k_b, gr.v_b};
      for (int sec = 0; sec < 3; sec++) {
        TnColsum cs; cs.dst[0] = bdst[sec]; cs.seg = bdst[sec] ? w : 0;
        if (defer) RC(wgrad(bf.d_qkv + sec * w, 3 * w, L.ln1, w, M, w, w, wdst[sec], cs));
        else RC(gemm_tn(bf.d_qkv + sec * w, 3 * w, L.ln1, w, M, w, w, wdst[sec], sc.wscratch, sc.wscratch_bytes, side.st, bdst[sec] ? &cs : nullptr));
      }
    }
  } else {
    // head-padded layout (head_dim < 64): sums and products land in the padded scratch, then the real columns are compacted
    TnColsum cs; cs.dst[0] = sc.gpad_b; cs.seg = 3 * wa;
    RC(gemm_tn(bf.d_qkv, 3 * wa, L.ln1, w, M, 3 * wa, w, sc.gpad, sc.wscratch, sc.wscratch_bytes, side.st, &cs));   // [3*H*64, w]
    float* dst[3] = {gr.q_w, gr.k_w, gr.v_w};
    float* bdst[3] = {gr.q_b, gr.k_b, gr.v_b};
    for (int sec = 0; sec < 3; sec++) {
      RC(map_copy_f32(dst[sec], sc.gpad + (size_t)sec * wa * w, ww, (long)d.Dh * w, (long)64 * w, side.st));
      if (bdst[sec]) RC(map_copy_f32(bdst[sec], sc.gpad_b + (size_t)sec * wa, (size_t)w, d.Dh, 64, side.st));
    }
  }
  {
    GemmEpi e; e.out_bf16 = sc.d_ln; e.ld_ob = w;
    RC(gemm_nt(bf.d_qkv, 3 * wa, L.wqkv_t, 3 * wa, M, w, 3 * wa, e, st));
  }
  // ---- LN1 (+ residual) over all rows: g = g2 + LNbwd(d_ln) ----
  RC(layernorm_bwd(sc.d_ln, x_in, L.mean1, L.rstd1, p.ln1_w, g2, g, bf.gb_out, gr.ln1_w, gr.ln1_b, M, w, 0, 0, 0, bf.ln1_part, bf.part_bytes, st, !defer));
  if (defer) {
    // the block's input gradient is done: everything its parameter gradients need exists.  One edge into the side stream, the two LayerNorm parameter
    // reductions, the grouped launch; "launch_no finished" is what lets block launch_no + 2 overwrite the ring entries this one read
    ForkJoin* fj = defer->fj;
    RC(stream_edge(fj, fj->wfork, st, fj->wside));
    RC(layernorm_bwd_reduce(bf.ln2_part, Mq, w, gr.ln2_w, gr.ln2_b, fj->wside));
    RC(layernorm_bwd_reduce(bf.ln1_part, M, w, gr.ln1_w, gr.ln1_b, fj->wside));
    // (the LAST block's launch - nothing runs beside it, the optimizer waits for it - runs every tile's token range as two halves on twice the CUs: the side
    // stream's slab scratch is free, nothing else of this backward uses it any more)
    RC(gemm_tn_grouped(jobs, nj, fj->wside, defer->last_block ? sc.wscratch : nullptr, sc.wscratch_bytes));
    RC(event_record_locked(fj, fj->wdone[launch_no & 3], fj->wside));
    if (dp_event) HIPRC(hipEventRecord(dp_event, fj->wside));
    return SMBV_OK;
  }
  // join: this block's parameter gradients are final on `st`, and every buffer the side products read (gb, d_act, gb2, d_qkv, the saved
  // activations) may be overwritten by the next block
  RC(side.end(st));
  if (dp_event) { if (pipe) RC(pipe->mark_inline(dp_event, st)); else HIPRC(hipEventRecord(dp_event, st)); }
  return SMBV_OK;
}
}  // namespace

int layer_bwd(const BlockPtrs& p, const BlockPtrs& gr, LayerWs& L, const BlockDims& d, BwdScratch& sc, const float* x_in, float* g, float* g2, WgradPipe& pipe,
              int Sq, int qoff, hipEvent_t dp_event, hipStream_t st) {
  const int k = pipe.k;
  RC(pipe.wait_for(k - 2, st));   // the ring entries this block overwrites were read by deferred launch k - 2
  BwdBufs bf;
  bf.g = g; bf.g2 = g2;
  bf.gb_in = pipe.gb[(2 * k) % 6]; bf.gb_mid = pipe.gb[(2 * k + 1) % 6]; bf.gb_out = pipe.gb[(2 * k + 2) % 6];
  bf.d_act = pipe.d_act[k & 1]; bf.d_qkv = pipe.d_qkv[k & 1];
  bf.ln1_part = pipe.ln1_part[k & 1]; bf.ln2_part = pipe.ln2_part[k & 1]; bf.part_bytes = pipe.part_bytes;
  const int w = d.w, ff = d.ff;
  const size_t ww = (size_t)w * w;
  const bool ranged = Sq > 0;
  const size_t Mmax = d.rows;
  const bool shapes_ok = w % 128 == 0 && ff % 128 == 0 && Mmax * (size_t)std::max(ff, 3 * w) * 2 < (1ull << 31);
  (void)ww;   // (q / k / v weight gradients that are not adjacent in the arena become three jobs of the grouped launch instead of one)
  const bool defer = pipe.fj && pipe.allow_defer && !(g_variant & 4) && !d.padded() && shapes_ok && !(ranged && d.B > 1) && !(g_variant & 32);
  // (Round 5 tried two other forms for the LAST block of a backward, whose deferred launch - ~200 us on 108 CUs - is what the optimizer waits for: every
  // product inline, and the MLP pair grouped beside the block's own attention backward with the projection / QKV products token-split at the end.  Both null
  // in the step, profiles/r05_tail_block_ab.txt: a 512-thread weight-gradient block beside 1,344 attention tiles is starved - 490-630 us for what takes 200 -
  // so the tail only moves from behind the last launch to behind the join.)
  const int rc = layer_bwd_core(p, gr, L, d, sc, x_in, bf, Sq, qoff, st, defer ? &pipe : nullptr, k, dp_event, &pipe);
  if (rc) return rc;
  pipe.deferred[k & 3] = defer;
  if (!defer) pipe.waited = k;   // an inline block joins the (in-order) side stream: everything before it has finished as well
  pipe.k = k + 1;
  return SMBV_OK;
}

}  // namespace smbv
