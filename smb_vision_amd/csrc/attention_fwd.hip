// Fused multi-head self-attention forward (flash-style, no mask, non-causal, dropout 0) for head_dim 64 and 32.
// Replaces eager_attention_forward / SDPA / flash-attn at src/models/videomae/modeling_videomae.py:196-223, 270-289:
//   O = softmax_fp32(Q K^T * Dh^-0.5) V, plus LSE per query row for the backward.
//
// Layout: Q, K, V are column slices of one token-major matrix [B*S, ld] (the fused QKV GEMM output), head h at
// columns h*Dh .. h*Dh+Dh-1; O is token-major [B*S, ldo].  No head-major copies are made.
//
// Work decomposition: grid (ceil(S/128), H, B); 4 waves per block, each wave owns 32 query rows; K/V tiles of 64 keys travel
// global -> LDS by LDS-DMA into a ring of three 16-KiB slots (two tiles in flight, one barrier per tile, no staging registers).
//   S^T[key][q]  = K * Q^T     v_mfma_f32_32x32x16_bf16, A = K fragment (ds_read_b128 from the swizzled K image),
//                              B = Q fragment (registers, loaded once) -> the query sits on the MFMA lane, so the row
//                              max / sum of the online softmax are lane-local (one cross-half exchange per tile).
//   O^T[d][q]   += V^T * P^T   the S^T accumulator, exponentiated and packed to bf16, IS the B operand (its rows are the
//                              contraction index); A = V^T fragment fetched with ds_read_b64_tr_b16 from the row-major
//                              V image -> no P round trip through LDS and no transposed copy of V.
// The loop is bound by VECTOR ISSUE at head_dim 64, not by the matrix pipe: per 64-key tile a wave issues 16 MFMAs (512 pipe cycles)
// and 32 v_exp (8 issue cycles each) + 32 adds + 16 converts + 16 v_max3 (MI355X_MICROARCH.md 'vector-instruction ISSUE cost').  So:
//   * Q is pre-multiplied by scale*log2(e) once: scores come out of the MFMA in the exp2 domain;
//   * the running reference -m sits in a persistent 16-register block used as the C input of the first QK MFMA of each
//     chain, so p = exp2(acc) needs no per-element subtract;
//   * NO per-tile maximum in the first sweep over the keys: with bf16 probabilities and fp32 sums the reference only has to keep exp2 finite, so
//     every tile is exponentiated against the first tile's row maxima (+ 32); a block whose row sums left the range repeats the sweep with guarded
//     tiles, in which the reference moves (and O, l are rescaled) when a tile's maximum exceeds it by more than 2^THR - see the kernel;
//   * no address arithmetic in the loop: the tile loop is unrolled over the ring's three slots, so every LDS read is one of six
//     per-lane base registers plus an immediate (with the slot computed from j, 44 of the loop's ~230 vector instructions were
//     address adds: 0.82 -> 0.70 ms at S = 20,480, H = 6);
//   * <= 168 registers -> THREE blocks per CU (three waves per SIMD) rather than hand-placing two waves' instruction streams: the
//     hand-phased two-wave form of round 1 measured 0.82 ms against 0.68 ms for this plain loop with guarded tiles, 0.63 ms with the optimistic
//     sweep (profiles/README.md).
// Softmax statistics in fp32.
#include "smbv_common.h"
#include "smbv_kernels.h"
#include <type_traits>

namespace smbv {

__device__ __forceinline__ int swz_k64(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// V image: 128-B rows, 32-B columns XOR-ed with bit1 of (key>>1) so that the 4 keys of a tr-read block hit distinct banks
__device__ __forceinline__ int swz_v64(int key, int c32) { return key * 128 + ((c32 ^ (((key >> 1) & 1) << 1)) << 5); }

static constexpr float ATTN_DEFER_THR = 6.0f;  // log2 units

// x of this lane and of lane l ^ 32 (v_permlane32_swap of x with itself: {own, other} below lane 32, {other, own} above; no LDS round trip)
__device__ __forceinline__ void both_halves(float x, float& u, float& v) {
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  const unsigned a = __builtin_bit_cast(unsigned, x);
  const u32x2v sw = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  unsigned lo = sw[0], hi = sw[1];
  asm volatile("" : "+v"(lo), "+v"(hi));  // (hipcc -O3 otherwise folds max(sw[0], sw[1]) to sw[0]: checked in the .s)
  u = __builtin_bit_cast(float, lo);
  v = __builtin_bit_cast(float, hi);
}
__device__ __forceinline__ float halves_max(float x) { float u, v; both_halves(x, u, v); return fmaxf(u, v); }
__device__ __forceinline__ float halves_sum(float x) { float u, v; both_halves(x, u, v); return u + v; }

__device__ __forceinline__ bf16x8 scale_bf16x8(u32x4 v, float c) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; i++) o[i] = pack_bf16(bf_lo(v[i]) * c, bf_hi(v[i]) * c);
  return __builtin_bit_cast(bf16x8, o);
}


// DH = 64: K / V images with 128-byte rows (swz_k64 / swz_v64), a copy piece is 8 rows.  DH = 32 (the V-JEPA predictor's heads): 64-byte rows,
// a copy piece is 16 rows; the K image keeps 16-B chunk c of row r at chunk c ^ ((r >> 2) & 3) (every ds_read_b128 lane group covers the 64 banks),
// the V image needs no swizzle at all (the 4 keys x 2 column halves x 4 sub-columns of a tr-read group already tile the 64 banks).
template <int DH>
__global__ __launch_bounds__(256, 3) void attn_fwd3_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, int ld,
                                                           bf16_t* __restrict__ O, bf16_t* __restrict__ Olo, int ldo, float* __restrict__ LSE, int S, int Sq, int qoff, int H, float scale, int guarded_only, int xcd_order) {
  static_assert(DH == 64 || DH == 32, "head widths with a native build");
  constexpr int KS = DH / 16, DT = DH / 32, RB = 2 * DH;  // QK^T k-steps, 32-channel output tiles, LDS row bytes
  constexpr int IMG = 64 * RB, SLOT = 2 * IMG;           // K image | V image
  constexpr int NP = IMG / 1024 / 4 * 2;                  // copy pieces per wave and tile (K + V): 4 at DH 64, 2 at DH 32
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * SLOT];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = l & 31, h2 = l >> 5;
  int tile_x = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  stagger_by_wave_slot((xcd_order >> 8) & 0xff);
  if (xcd_order & 1) xcd_block_order((Sq + 127) / 128, H, tile_x, head, b);
  const size_t row0 = (size_t)b * S;
  const int q = tile_x * 128 + w * 32 + r;
  const int hoff = head * DH;
  const float c = scale * 1.4426950408889634f;

  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    u32x4 v = {0, 0, 0, 0};
    if (q < Sq) v = *(const u32x4*)(Q + (row0 + qoff + q) * ld + hoff + 16 * ks + 8 * h2);
    asm volatile("" : "+v"(v));  // consumed here: the wait for these loads stays out of the tile loop
    qf[ks] = scale_bf16x8(v, c);
  }
  f32x16 ot[DT], cneg;
#pragma unroll
  for (int i = 0; i < 16; i++) {
#pragma unroll
    for (int dt = 0; dt < DT; dt++) ot[dt][i] = 0.f;
    cneg[i] = 0.f;
  }
  float m2 = 0.f, l_a = 0.f;
  float ref_shift = 0.f;  // added to the first tile's row maxima when they become the reference (the optimistic sweep, below)

  const i32x4 rsk = dma_rsrc(K + row0 * ld + hoff, (unsigned)(((size_t)(S - 1) * ld + DH) * 2));
  const i32x4 rsv = dma_rsrc(V + row0 * ld + hoff, (unsigned)(((size_t)(S - 1) * ld + DH) * 2));
  // copy side (the LDS end of an LDS-DMA is lane-linear, so the image's chunk permutation is applied to the SOURCE address)
  int ok0, ok1 = 0, ov0, ov1 = 0;
  if constexpr (DH == 64) {  // wave w: rows 16 w .. + 15 of both images as two pieces of 8 rows
    const int rw0 = 16 * w + (l >> 3), rw1 = rw0 + 8;
    ok0 = (rw0 * ld + ((l & 7) ^ ((rw0 >> 1) & 7)) * 8) * 2;
    ok1 = (rw1 * ld + ((l & 7) ^ ((rw1 >> 1) & 7)) * 8) * 2;
    ov0 = (rw0 * ld + ((l & 7) ^ (((rw0 >> 1) & 1) << 2)) * 8) * 2;
    ov1 = (rw1 * ld + ((l & 7) ^ (((rw1 >> 1) & 1) << 2)) * 8) * 2;
  } else {                   // wave w: rows 16 w .. + 15 of both images as one piece of 16 rows
    const int rw = 16 * w + (l >> 2);
    ok0 = (rw * ld + ((l & 3) ^ ((rw >> 2) & 3)) * 8) * 2;
    ov0 = (rw * ld + (l & 3) * 8) * 2;
  }
  const int tile_bytes = 64 * ld * 2;
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem) + w * (IMG / 4)));
  auto stage = [&](int j, int slot) __attribute__((always_inline)) {
    const unsigned dk = lds_w + (unsigned)slot * SLOT, dv = dk + IMG;
    const int ko = j * tile_bytes;
    dma16_hidden(rsk, dk, ok0 + ko);
    if constexpr (DH == 64) dma16_hidden(rsk, dk + 1024, ok1 + ko);
    dma16_hidden(rsv, dv, ov0 + ko);
    if constexpr (DH == 64) dma16_hidden(rsv, dv + 1024, ov1 + ko);
  };
  const int g1 = (l >> 4) & 1, qq = (l >> 2) & 3, pp = l & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // per-lane LDS offsets; everything else about a read (slot, key tile, k-step) is an immediate.  DH 64: swz_k64(32 kt2 + r, 2 ks + h2) = ka[ks] +
  // 4096 kt2,  swz_v64(16 kstep + 8 e + 4 h2 + qq, 2 dt + g1) + 8 pp = va[dt] + 2048 kstep + 1024 e  (bit 1 of the key is bit 1 of qq);  DH 32: the
  // same with 64-byte rows
  const unsigned char* ka[KS];
  const unsigned char* va[DT];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) ka[ks] = smem + (DH == 64 ? swz_k64(r, 2 * ks + h2) : r * 64 + (((2 * ks + h2) ^ ((r >> 2) & 3)) << 4));
#pragma unroll
  for (int dt = 0; dt < DT; dt++) va[dt] = smem + IMG + (DH == 64 ? swz_v64(4 * h2 + qq, 2 * dt + g1) : (4 * h2 + qq) * 64 + 32 * g1) + 8 * pp;

  // slot: a literal at each call site.  guard (std::true_type / std::false_type): with it, the tile's row maximum is formed and the running reference
  // raised when a score outgrows it by 2^THR; without it, the scores are exponentiated against the reference as it stands (see the sweeps below).
  auto tile = [&](int j, int slot, auto guard) __attribute__((always_inline)) {
    constexpr bool GUARD = decltype(guard)::value;
    f32x16 sc[2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; kt2++) {
      bf16x8 kf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ks++) kf[ks] = *(const bf16x8*)(ka[ks] + slot * SLOT + 32 * RB * kt2);
#pragma unroll
      for (int ks = 0; ks < KS; ks++) sc[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ks == 0 ? cneg : sc[kt2], 0, 0, 0);   // S'^T - m2
    }
    if (j * 64 + 64 > S) {  // ragged last tile: keys >= S get -inf
#pragma unroll
      for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int key = j * 64 + 32 * kt2 + (i & 3) + 8 * (i >> 2) + 4 * h2;
          if (key >= S) sc[kt2][i] = -INFINITY;
        }
    }
    if constexpr (GUARD) {
      float mx = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
      for (int i = 1; i < 16; i++) mx = fmaxf(fmaxf(mx, sc[0][i]), sc[1][i]);
      mx = halves_max(mx);
      const bool rebase = j == 0 || mx > ATTN_DEFER_THR;
      if (__builtin_amdgcn_ballot_w64(rebase) != 0) {  // rare after the first tile: the reference only moves when a score outgrows it by 2^6
        const float d = rebase ? mx + (j == 0 ? ref_shift : 0.f) : 0.f;
        const float alpha = j == 0 ? 0.f : fast_exp2(-d);  // (nothing is accumulated before tile 0: no 0 * inf for very negative first maxima)
        m2 += d;
        l_a *= alpha;
#pragma unroll
        for (int i = 0; i < 16; i++) {
#pragma unroll
          for (int dt = 0; dt < DT; dt++) ot[dt][i] *= alpha;
          sc[0][i] -= d;
          sc[1][i] -= d;
          cneg[i] = -m2;
        }
      }
    }
#pragma unroll
    for (int kstep = 0; kstep < 4; kstep++) {
      bf16x8 pf;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float p = fast_exp2(sc[kstep >> 1][8 * (kstep & 1) + i]);
        l_a += p;
        pf[i] = (__bf16)p;
      }
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const s16x4 v0 = lds_read_tr16(va[dt] + slot * SLOT + 16 * RB * kstep);
        const s16x4 v1 = lds_read_tr16(va[dt] + slot * SLOT + 16 * RB * kstep + 8 * RB);
        const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, ot[dt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile j have returned before it can reach the next barrier
  };
  auto landed = [&]() __attribute__((always_inline)) {  // this wave's NP pieces of the oldest tile in flight (the next tile's NP may still fly)
    if constexpr (NP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the tile is published; everybody is done with the tile before it, whose slot the next copy takes
  };
  const int nt = (S + 63) / 64;
  // One sweep over the key tiles (tile j lives in ring slot j % 3; the first tile always sets the reference to its row maxima).
  auto sweep = [&](auto guard) __attribute__((always_inline)) {
    stage(0, 0);
    stage(1, 1);
    landed();
    stage(2, 2);
    tile(0, 0, std::true_type{});
    for (int j = 1; j < nt; j += 3) {
      landed();
      stage(j + 2, 0);
      tile(j, 1, guard);
      if (j + 1 >= nt) break;
      landed();
      stage(j + 3, 1);
      tile(j + 1, 2, guard);
      if (j + 2 >= nt) break;
      landed();
      stage(j + 4, 2);
      tile(j + 2, 0, guard);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the sweep (or the workgroup)
  };
  // OPTIMISTIC sweep first: with bf16 probabilities and fp32 sums the reference only has to keep exp2(s - m) finite - accuracy does not depend on it - so the
  // sweep exponentiates every later tile against the FIRST tile's row maxima and forms no maximum at all (21 of the loop's ~190 vector instructions per tile,
  // 6-9 % of the kernel: the loop is vector-issue bound).  It is valid iff no row sum left the range: a score more than ~100 (log2 units) above its row's
  // first-tile maximum shows as a sum >= 2^100 / inf / NaN.  Then - decided for the whole block, whose waves share the K / V stream - everything is reset and
  // the sweep repeated with the guarded tiles.  Scores FAR BELOW the reference underflow to 0, which is what their weight is.
  // (its reference sits 2^32 ABOVE the first tile's maxima: what can overflow is a later score, so the window is put where it helps - scores up to ~130
  // log2 units above the first tile's maximum stay in range, scores more than ~94 below it flush to the zero their weight rounds to anyway)
  if (guarded_only) sweep(std::true_type{});  // debug bit 2048 (A/B and the equivalence test)
  else { ref_shift = 32.f; sweep(std::false_type{}); ref_shift = 0.f; }
  float l_tot = halves_sum(l_a);
  if (!guarded_only) {
    int* flag = (int*)smem;
    __syncthreads();  // every wave is done with the ring
    if (t == 0) *flag = 0;
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(!(l_tot < 0x1p100f)) != 0 && l == 0) *(volatile int*)flag = 1;
    __syncthreads();
    const int redo = *(volatile int*)flag;
    __syncthreads();  // (the flag lives in the ring the next sweep overwrites)
    if (redo) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
#pragma unroll
        for (int dt = 0; dt < DT; dt++) ot[dt][i] = 0.f;
        cneg[i] = 0.f;
      }
      m2 = 0.f;
      l_a = 0.f;
      sweep(std::true_type{});
      l_tot = halves_sum(l_a);
    }
  }
  const float inv = 1.0f / l_tot;
  if (q < Sq) {
    bf16_t* op = O + (row0 + qoff + q) * ldo + hoff;
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int d0 = 32 * dt + 8 * rg + 4 * h2;
        const float o0 = ot[dt][4 * rg] * inv, o1 = ot[dt][4 * rg + 1] * inv, o2 = ot[dt][4 * rg + 2] * inv, o3 = ot[dt][4 * rg + 3] * inv;
        const u32x2 pk = {pack_bf16(o0, o1), pack_bf16(o2, o3)};
        *(u32x2*)(op + d0) = pk;
        if (Olo) {  // the bf16 residual of the fp32 output: O_hi + O_lo carries 16 significant bits into delta = rowsum(dO * O) (attention_bwd.hip)
          const u32x2 lo = {pack_bf16(o0 - bf_lo(pk[0]), o1 - bf_hi(pk[0])), pack_bf16(o2 - bf_lo(pk[1]), o3 - bf_hi(pk[1]))};
          *(u32x2*)(Olo + (row0 + qoff + q) * ldo + hoff + d0) = lo;
        }
      }
    if (h2 == 0) LSE[((size_t)b * H + head) * Sq + q] = (m2 + __log2f(l_tot)) * 0.6931471805599453f;
  }
}

extern int g_variant;  // debug hook, capi.hip (smbv_debug_set_variant)

int attention_fwd(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, bf16_t* O, int ldo, float* LSE, int B, int S, int H, int Dh,
                  float scale, int Sq, int qoff, hipStream_t st, bf16_t* O_lo) {
  if (Sq <= 0) { Sq = S; qoff = 0; }
  if (qoff < 0 || qoff + Sq > S) { set_error("attention_fwd: query range [%d, %d) outside the sequence (%d)", qoff, qoff + Sq, S); return SMBV_ERR_SHAPE; }
  if (Dh != 64 && Dh != 32) { set_error("attention_fwd: head_dim %d not supported (64 or 32; other widths run zero-padded in 64-wide slots, block.h)", Dh); return SMBV_ERR_SHAPE; }
  if (B <= 0 || S <= 0 || H <= 0 || (ld & 7) || (ldo & 3)) { set_error("attention_fwd: bad shape B=%d S=%d H=%d ld=%d ldo=%d", B, S, H, ld, ldo); return SMBV_ERR_SHAPE; }
  if ((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) || (((uintptr_t)O | (uintptr_t)O_lo) & 7)) { set_error("attention_fwd: alignment"); return SMBV_ERR_ALIGN; }
  if ((size_t)(S + 512) * ld * 2 >= (1ull << 31)) { set_error("attention_fwd: one sequence spans %zu bytes of the QKV matrix; the LDS-DMA offsets are 32-bit", (size_t)S * ld * 2); return SMBV_ERR_SHAPE; }
  // start-up stagger by SIMD wave slot (smbv_common.h): 2 x 512 clocks per slot; -1 % (head_dim 64) .. -3 % (head_dim 32) for the forward, null for the backward
  // kernels (profiles/r03_attn_stagger_ab.txt); bits 26-28 override the units, bit 29 switches it off
  const int stag = (g_variant & (1 << 29)) ? 0 : (((g_variant >> 26) & 7) ? ((g_variant >> 26) & 7) : 2);
  const int xo = ((g_variant & (1 << 24)) ? 0 : 1) | (stag << 8);
  const bool xgrid = xo & 1;  // XCD-aware block order (smbv_common.h xcd_block_order): -1..3 % at every shape (tools/attn_ab.py); bit 24 = plain grid
  const dim3 grid = xgrid ? dim3(cdiv(Sq, 128) * H * B) : dim3(cdiv(Sq, 128), H, B);
  if (Dh == 64) hipLaunchKernelGGL(attn_fwd3_kernel<64>, grid, dim3(256), 0, st, Q, K, V, ld, O, O_lo, ldo, LSE, S, Sq, qoff, H, scale, (g_variant & 2048) ? 1 : 0, xo);
  else hipLaunchKernelGGL(attn_fwd3_kernel<32>, grid, dim3(256), 0, st, Q, K, V, ld, O, O_lo, ldo, LSE, S, Sq, qoff, H, scale, (g_variant & 2048) ? 1 : 0, xo);
  return check_launch("attention_fwd");
}

}  // namespace smbv
