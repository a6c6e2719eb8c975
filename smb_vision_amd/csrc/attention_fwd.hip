// Fused multi-head self-attention forward (flash-style, no mask, non-causal, dropout 0) for head_dim 64.
// Replaces eager_attention_forward / SDPA / flash-attn at src/models/videomae/modeling_videomae.py:196-223, 270-289:
//   O = softmax_fp32(Q K^T * Dh^-0.5) V, plus LSE per query row for the backward.
//
// Layout: Q, K, V are column slices of one token-major matrix [B*S, ld] (the fused QKV GEMM output), head h at
// columns h*64 .. h*64+63; O is token-major [B*S, ldo].  No head-major copies are made.
//
// Work decomposition: grid (ceil(S/128), H, B); 4 waves per block, each wave owns 32 query rows; K/V tiles of 64
// keys staged global -> registers -> LDS (double-buffered, one barrier per tile).
//   S^T[key][q]  = K * Q^T     v_mfma_f32_32x32x16_bf16, A = K fragment (ds_read_b128 from the swizzled K image),
//                              B = Q fragment (registers, loaded once) -> the query sits on the MFMA lane, so the row
//                              max / sum of the online softmax are lane-local (one cross-half shuffle per tile).
//   O^T[d][q]   += V^T * P^T   the S^T accumulator, exponentiated and packed to bf16, IS the B operand (its rows are the
//                              contraction index); A = V^T fragment fetched with ds_read_b64_tr_b16 from the row-major
//                              V image -> no P round trip through LDS and no transposed copy of V.
// VALU diet (the loop is VALU-issue bound at head_dim 64, not MFMA bound):
//   * Q is pre-multiplied by scale*log2(e) once, so scores come out of the MFMA in the exp2 domain;
//   * the running reference -m sits in a persistent 16-register block used as the C input of the first QK MFMA of each
//     chain, so p = exp2(acc) needs no per-element subtract;
//   * the reference is only moved (and O, l rescaled) when a tile's maximum exceeds it by more than 2^THR (deferred
//     max): exp2 arguments stay <= THR, exact in fp32; tile 0 always rebases.
// Softmax statistics in fp32.
#include "smbv_common.h"
#include "smbv_kernels.h"

namespace smbv {

__device__ __forceinline__ int swz_k64(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// V image: 128-B rows, 32-B columns XOR-ed with bit1 of (key>>1) so that the 4 keys of a tr-read block hit distinct banks
__device__ __forceinline__ int swz_v64(int key, int c32) { return key * 128 + ((c32 ^ (((key >> 1) & 1) << 1)) << 5); }

static constexpr float ATTN_DEFER_THR = 6.0f;  // log2 units

__device__ __forceinline__ bf16x8 scale_bf16x8(u32x4 v, float c) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; i++) o[i] = pack_bf16(bf_lo(v[i]) * c, bf_hi(v[i]) * c);
  return __builtin_bit_cast(bf16x8, o);
}


#define SG_VALU 0x2
#define SG_MFMA 0x8
#define SG_DSR 0x100
#define SG_TRANS 0x400

// Hand-phased, rotated forward: while the matrix pipe runs the QK^T chains of tile j+1 and the PV products of tile j, the wave's
// VALU slots carry the exp2 / row-sum / bf16-convert work of tile j and the row maximum of tile j+1, and every LDS fragment is read
// at least four MFMAs before its consumer.  Per 64-key tile, four groups of 4 MFMAs:
//   G1  S'^T(j+1)[keys 0..31]   | P(j) k-step 0 (8 exp2, 8 adds, 4 converts) | reads: all V^T fragments of tile j
//   G2  S'^T(j+1)[keys 32..63]  | P(j) k-step 1
//   -- barrier: tile j is dead -> its ring slot takes tile j+4 (LDS-DMA); tile j+2 has landed (counted vmcnt) --
//   G3  O^T += V^T P^T, k-steps 0,1 | P(j) k-step 2                          | reads: K rows 0..31 of tile j+2
//   G4  O^T += V^T P^T, k-steps 2,3 | P(j) k-step 3, then max of S'(j+1)     | reads: K rows 32..63 of tile j+2
// followed by the (rare, wave-uniform) rebase of the running reference.  K/V travel global -> LDS by LDS-DMA into a ring of four
// 64-key slots, two tiles of flight time, no staging registers.  Arithmetic is that of attn_fwd_kernel except for the order in
// which a tile's 64 probabilities enter the row sum.
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                           const bf16_t* __restrict__ V, int ld, bf16_t* __restrict__ O, int ldo,
                                                           float* __restrict__ LSE, int S, int Sq, int qoff, int H, float scale) {
  constexpr int SLOT = 2 * 64 * 128;  // K image | V image
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * SLOT];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int r = l & 31, h2 = l >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const size_t row0 = (size_t)b * S;
  const int q = blockIdx.x * 128 + w * 32 + r;
  const int hoff = head * 64;
  const float c = scale * 1.4426950408889634f;

  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
    u32x4 v = {0, 0, 0, 0};
    if (q < Sq) v = *(const u32x4*)(Q + (row0 + qoff + q) * ld + hoff + 16 * ks + 8 * h2);
    qf[ks] = scale_bf16x8(v, c);
  }
  f32x16 ot[2], cneg;
#pragma unroll
  for (int i = 0; i < 16; i++) { ot[0][i] = 0.f; ot[1][i] = 0.f; cneg[i] = 0.f; }
  float m2 = 0.f, l_a = 0.f, l_b = 0.f;

  // LDS-DMA: wave w copies rows 16w..16w+15 of the K and the V tile (2 + 2 pieces of 8 rows x 128 B), image swizzles applied to the
  // source chunk; rows >= S lie outside the descriptors and arrive as zeros, so tiles past the end may be requested freely.
  const i32x4 rsk = dma_rsrc(K + row0 * ld + hoff, (unsigned)(((size_t)(S - 1) * ld + 64) * 2));
  const i32x4 rsv = dma_rsrc(V + row0 * ld + hoff, (unsigned)(((size_t)(S - 1) * ld + 64) * 2));
  const int rw0 = 16 * w + (l >> 3), rw1 = rw0 + 8;
  const int ok0 = (rw0 * ld + ((l & 7) ^ ((rw0 >> 1) & 7)) * 8) * 2, ok1 = (rw1 * ld + ((l & 7) ^ ((rw1 >> 1) & 7)) * 8) * 2;
  const int ov0 = (rw0 * ld + ((l & 7) ^ (((rw0 >> 1) & 1) << 2)) * 8) * 2, ov1 = (rw1 * ld + ((l & 7) ^ (((rw1 >> 1) & 1) << 2)) * 8) * 2;
  const int tile_bytes = 64 * ld * 2;
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem) + w * 2048));
  auto stage = [&](int j) __attribute__((always_inline)) {
    const unsigned dk = lds_w + (unsigned)(j & 3) * SLOT, dv = dk + 64 * 128;
    const int ko = j * tile_bytes;
    dma16_hidden(rsk, dk, ok0 + ko);
    dma16_hidden(rsk, dk + 1024, ok1 + ko);
    dma16_hidden(rsv, dv, ov0 + ko);
    dma16_hidden(rsv, dv + 1024, ov1 + ko);
  };
  const int g1 = (l >> 4) & 1, qq = (l >> 2) & 3, pp = l & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto kread = [&](int j, int kt2, bf16x8* kf) __attribute__((always_inline)) {
    const unsigned char* sk = smem + (j & 3) * SLOT;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) kf[ks] = *(const bf16x8*)(sk + swz_k64(32 * kt2 + r, 2 * ks + h2));
  };
  auto vread = [&](int j, int kstep, bf16x8* vt) __attribute__((always_inline)) {  // vt[dt]
    const unsigned char* sv = smem + (j & 3) * SLOT + 64 * 128;
    const int kb = 16 * kstep + 4 * h2 + qq;
#pragma unroll
    for (int dt = 0; dt < 2; dt++) {
      const int c32 = 2 * dt + g1;
      const s16x4 v0 = lds_read_tr16(sv + swz_v64(kb, c32) + 8 * pp);
      const s16x4 v1 = lds_read_tr16(sv + swz_v64(kb + 8, c32) + 8 * pp);
      const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      vt[dt] = __builtin_bit_cast(bf16x8, vv);
    }
  };
  auto pgroup = [&](f32x16& sb, int s2, bf16x8& pf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const float p0 = fast_exp2(sb[8 * s2 + i]), p1 = fast_exp2(sb[8 * s2 + i + 1]);
      l_a += p0;
      l_b += p1;
      pf[i] = (__bf16)p0;
      pf[i + 1] = (__bf16)p1;
    }
    // (pinned: hipcc otherwise sinks the row-sum adds - nothing in the tile needs them - below the tile's last MFMA, where they run
    // with the matrix pipe idle; row sums as a fifth MFMA product (ones * P^T) were tried too: same speed, the bound just moves)
    asm volatile("" : "+v"(l_a), "+v"(l_b));
  };
  const int nt = (S + 63) / 64;
  auto tile_max = [&](const f32x16* sn) __attribute__((always_inline)) -> float {
    float mx = sn[0][0];
#pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, sn[0][i]);
#pragma unroll
    for (int i = 0; i < 16; i++) mx = fmaxf(mx, sn[1][i]);
    return mx;
  };
  // tile j's scores (relative to the running reference) are complete and mx is their per-lane maximum: mask a ragged last tile,
  // combine the two lane halves and, when the maximum exceeds the threshold, move the reference (rare, wave-uniform)
  auto settle = [&](f32x16* sn, float mx, int j, bool first) __attribute__((always_inline)) {
    if (j * 64 + 64 > S && j * 64 < S) {  // ragged last tile: keys >= S get -inf
#pragma unroll
      for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int key = j * 64 + 32 * kt2 + (i & 3) + 8 * (i >> 2) + 4 * h2;
          if (key >= S) sn[kt2][i] = -INFINITY;
        }
      mx = tile_max(sn);
    }
    {  // the other lane half holds the other 32 keys of the same query row
      typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
      const unsigned ma = __builtin_bit_cast(unsigned, mx);
      const u32x2v sw = __builtin_amdgcn_permlane32_swap(ma, ma, false, false);
      unsigned lo = sw[0], hi = sw[1];
      asm volatile("" : "+v"(lo), "+v"(hi));  // (hipcc -O3 otherwise folds max(sw[0], sw[1]) to sw[0]: checked in the .s)
      mx = fmaxf(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
    }
    const bool rebase = first || (j < nt && mx > ATTN_DEFER_THR);
    if (__builtin_amdgcn_ballot_w64(rebase) != 0) {
      const float d = rebase ? mx : 0.f;
      const float alpha = fast_exp2(-d);
      m2 += d;
      l_a *= alpha;
      l_b *= alpha;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        ot[0][i] *= alpha;
        ot[1][i] *= alpha;
        sn[0][i] -= d;
        sn[1][i] -= d;
        cneg[i] = -m2;
      }
    }
  };

  stage(0); stage(1); stage(2); stage(3);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tiles 0 and 1
  __builtin_amdgcn_s_barrier();
  bf16x8 kfa[4], kfb[4];
  f32x16 sA[2], sB[2];
  kread(0, 0, kfa);
  kread(0, 1, kfb);
#pragma unroll
  for (int ks = 0; ks < 4; ks++) sA[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[ks], qf[ks], ks == 0 ? cneg : sA[0], 0, 0, 0);
#pragma unroll
  for (int ks = 0; ks < 4; ks++) sA[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[ks], qf[ks], ks == 0 ? cneg : sA[1], 0, 0, 0);
  kread(1, 0, kfa);
  kread(1, 1, kfb);
  settle(sA, tile_max(sA), 0, true);

  // one pipeline body: finishes tile j (scores in sc) and starts tile j+1 (K fragments in kfa/kfb) into sn
  auto body = [&](f32x16* sc, f32x16* sn, int j) __attribute__((always_inline)) {
    bf16x8 pf[4], vt[4][2];
    // ---- G1
#pragma unroll
    for (int ks = 0; ks < 4; ks++) sn[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[ks], qf[ks], ks == 0 ? cneg : sn[0], 0, 0, 0);
    pgroup(sc[0], 0, pf[0]);
    vread(j, 0, vt[0]);
    vread(j, 1, vt[1]);
    vread(j, 2, vt[2]);  // all of tile j's V^T fragments are requested here, a whole MFMA chain ahead of the hand-over barrier:
    vread(j, 3, vt[3]);  // the lgkmcnt(0) in front of it then costs nothing
#pragma unroll
    for (int i = 0; i < 4; i++) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 2, 0);
      __builtin_amdgcn_sched_group_barrier(SG_VALU, 3, 0);
      __builtin_amdgcn_sched_group_barrier(SG_DSR, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- G2
#pragma unroll
    for (int ks = 0; ks < 4; ks++) sn[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[ks], qf[ks], ks == 0 ? cneg : sn[1], 0, 0, 0);
    pgroup(sc[0], 1, pf[1]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 2, 0);
      __builtin_amdgcn_sched_group_barrier(SG_VALU, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- every wave is past its last read of tile j; this wave's pieces of tile j+2 have landed (tile j+3 may still fly)
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");  // (lgkmcnt: this wave's reads of tile j have returned before its slot is handed back)
    __builtin_amdgcn_s_barrier();
    stage(j + 4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- G3
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int dt = 0; dt < 2; dt++) ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt[ks][dt], pf[ks], ot[dt], 0, 0, 0);
    pgroup(sc[1], 0, pf[2]);
    kread(j + 2, 0, kfa);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 2, 0);
      __builtin_amdgcn_sched_group_barrier(SG_VALU, 3, 0);
      __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- G4
    pgroup(sc[1], 1, pf[3]);
#pragma unroll
    for (int ks = 2; ks < 4; ks++)
#pragma unroll
      for (int dt = 0; dt < 2; dt++) ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt[ks][dt], pf[ks], ot[dt], 0, 0, 0);
    kread(j + 2, 1, kfb);
    float mx = tile_max(sn);
    asm volatile("" : "+v"(mx));  // (pinned inside this group: see pgroup)
    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
    __builtin_amdgcn_sched_group_barrier(SG_TRANS, 4, 0);
    __builtin_amdgcn_sched_group_barrier(SG_VALU, 6, 0);
    __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
    __builtin_amdgcn_sched_group_barrier(SG_TRANS, 4, 0);
    __builtin_amdgcn_sched_group_barrier(SG_VALU, 6, 0);
    __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
    __builtin_amdgcn_sched_group_barrier(SG_VALU, 8, 0);
    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
    __builtin_amdgcn_sched_group_barrier(SG_VALU, 9, 0);
    __builtin_amdgcn_sched_barrier(0);
    settle(sn, mx, j + 1, false);
  };
  for (int j = 0; j < nt; j += 2) {
    body(sA, sB, j);
    if (j + 1 < nt) body(sB, sA, j + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup

  const float l_i = l_a + l_b;
  const float l_tot = l_i + __shfl_xor(l_i, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q < Sq) {
    bf16_t* op = O + (row0 + qoff + q) * ldo + hoff;
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int d0 = 32 * dt + 8 * rg + 4 * h2;
        u32x2 pk = {pack_bf16(ot[dt][4 * rg] * inv, ot[dt][4 * rg + 1] * inv), pack_bf16(ot[dt][4 * rg + 2] * inv, ot[dt][4 * rg + 3] * inv)};
        *(u32x2*)(op + d0) = pk;
      }
    if (h2 == 0) LSE[((size_t)b * H + head) * Sq + q] = (m2 + __log2f(l_tot)) * 0.6931471805599453f;
  }
}

extern int g_variant;  // debug hook, capi.hip (smbv_debug_set_variant)

int attention_fwd(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, bf16_t* O, int ldo, float* LSE, int B, int S, int H, int Dh,
                  float scale, int Sq, int qoff, hipStream_t st) {
  if (Sq <= 0) { Sq = S; qoff = 0; }
  if (qoff < 0 || qoff + Sq > S) { set_error("attention_fwd: query range [%d, %d) outside the sequence (%d)", qoff, qoff + Sq, S); return SMBV_ERR_SHAPE; }
  if (Dh != 64) { set_error("attention_fwd: head_dim %d not supported (only 64)", Dh); return SMBV_ERR_SHAPE; }
  if (B <= 0 || S <= 0 || H <= 0 || (ld & 7) || (ldo & 3)) { set_error("attention_fwd: bad shape B=%d S=%d H=%d ld=%d ldo=%d", B, S, H, ld, ldo); return SMBV_ERR_SHAPE; }
  if ((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) || ((uintptr_t)O & 7)) { set_error("attention_fwd: alignment"); return SMBV_ERR_ALIGN; }
  if ((size_t)(S + 256) * ld * 2 >= (1ull << 31)) { set_error("attention_fwd: one sequence spans %zu bytes of the QKV matrix; the LDS-DMA offsets are 32-bit", (size_t)S * ld * 2); return SMBV_ERR_SHAPE; }
  hipLaunchKernelGGL(attn_fwd2_kernel, dim3(cdiv(Sq, 128), H, B), dim3(256), 0, st, Q, K, V, ld, O, ldo, LSE, S, Sq, qoff, H, scale);
  return check_launch("attention_fwd");
}

}  // namespace smbv
