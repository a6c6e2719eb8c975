// Fused multi-head self-attention backward (flash-style recompute from LSE), head_dim 64 and 32 (template parameter DH).
// Autograd of src/models/videomae/modeling_videomae.py:196-223 (softmax(QK^T*scale) V).
//
// No float atomics (bitwise reproducible).  Two kinds of blocks, launched together as ONE grid (attn_bwd_pair_kernel, below):
//   key-owner    (attn_dkv2_block) a block owns 128 keys (4 waves x 32 keys, key on the MFMA lane), sweeps the query tiles;
//                S = Q K^T and dP = dO V^T come out with the key on the lane and the query in the registers, so P and dS
//                are directly the B operands of dV^T += dO^T P and dK^T += Q^T dS (A operands by ds_read_b64_tr_b16
//                from the same row-major Q / dO LDS image the row reads use).
//   query-owner  (attn_dq2_block) a block owns 128 queries (query on the lane), sweeps the key tiles;
//                S^T = K Q^T, dP^T = V dO^T, dS^T is the B operand of dQ^T += K^T dS^T (A operand by tr-read of K).
// delta[q] = sum_d dO[q,d] * O[q,d] comes from the out-projection's input-gradient epilogue in the block backward (gemm.hip) or from
// attn_delta_kernel (parts bit 0).  The split costs 7 MFMA products instead of 5 but needs no cross-workgroup reduction at all (a dQ
// accumulated with float atomics would be bound by the ~1.3 TB/s atomic rate at S = 20,480; an ordered hand-off needs every key block
// of a head resident at once, which at head_dim 64 does not fit the register file: DESIGN.md section 4).
#include "smbv_common.h"
#include "smbv_kernels.h"
#include "block.h"
#include <mutex>
#include <cstring>
#include <utility>
#include <type_traits>

namespace smbv {
extern int g_variant;
extern int g_ab;   // SMBV_DEBUG_AB (capi.hip): bit 8 = the rotary map of dQ / dK as a launch of its own


// 128-byte-row LDS image usable for ds_read_b128 row reads (16 rows x same chunk) AND ds_read_b64_tr_b16 blocks
// (4 consecutive rows x 32 B): chunk16 ^= f((row>>1)&7), f(x) = ((x&1)<<2) | (x>>1).
__device__ __forceinline__ int swz128(int row, int c16) {
  const int x = (row >> 1) & 7;
  return row * 128 + ((c16 ^ (((x & 1) << 2) | (x >> 1))) << 4);
}

// image row swizzle by head width: 64 -> swz128; 32 -> 64-byte rows, 16-B chunk c of row r at chunk c ^ ((r >> 2) & 3): every ds_read_b128 lane
// group covers the 64 banks, and within a ds_read_b64_tr_b16 group (4 consecutive rows of a 4-aligned block) the XOR is a constant, so the 4 rows x
// 4 chunks x 2 halves tile the banks as well
template <int DH>
__device__ __forceinline__ int swzr(int row, int c16) {
  if constexpr (DH == 64) return swz128(row, c16);
  else return row * 64 + ((c16 ^ ((row >> 2) & 3)) << 4);
}
// per-lane SOURCE byte offset (row-major operand, leading dimension ld elements) of LDS-DMA piece i of a wave's 32 rows: a piece lands lane-linear
// (1 KiB: 8 rows x 128 B at DH 64, 16 rows x 64 B at DH 32), so the image's chunk permutation is applied to the source chunk
template <int DH>
__device__ __forceinline__ int dma_src_off(int w, int i, int l, int ld) {
  if constexpr (DH == 64) {
    const int row = 32 * w + 8 * i + (l >> 3);
    const int x = (row >> 1) & 7;
    return (row * ld + ((l & 7) ^ (((x & 1) << 2) | (x >> 1))) * 8) * 2;
  } else {
    const int row = 32 * w + 16 * i + (l >> 2);
    return (row * ld + ((l & 3) ^ ((row >> 2) & 3)) * 8) * 2;
  }
}

__device__ __forceinline__ bf16x8 scale_frag(u32x4 v, float c) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; i++) o[i] = pack_bf16(bf_lo(v[i]) * c, bf_hi(v[i]) * c);
  return __builtin_bit_cast(bf16x8, o);
}

template <int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ O, const bf16_t* __restrict__ Olo, const bf16_t* __restrict__ dO, int ldo,
                                                         float* __restrict__ delta, int B, int S, int Sq, int qoff, int H) {
  // DH / 8 lanes per (row, head): 16 B each
  constexpr int LP = DH / 8;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long pair = gid / LP;
  const int sub = gid % LP;
  const long total = (long)B * Sq * H;
  float acc = 0.f;
  if (pair < total) {
    const long rq = pair / H;
    const long row = (rq / Sq) * S + qoff + (rq % Sq);
    const int head = pair % H;
    const u32x4 o = *(const u32x4*)(O + row * ldo + head * DH + sub * 8);
    const u32x4 g = *(const u32x4*)(dO + row * ldo + head * DH + sub * 8);
    if (Olo) {  // O = hi + lo (the forward's bf16 residual of its fp32 output)
      const u32x4 ol = *(const u32x4*)(Olo + row * ldo + head * DH + sub * 8);
#pragma unroll
      for (int i = 0; i < 4; i++) acc += (bf_lo(o[i]) + bf_lo(ol[i])) * bf_lo(g[i]) + (bf_hi(o[i]) + bf_hi(ol[i])) * bf_hi(g[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) acc += bf_lo(o[i]) * bf_lo(g[i]) + bf_hi(o[i]) * bf_hi(g[i]);
    }
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  if constexpr (LP == 8) acc += __shfl_xor(acc, 4, 64);
  if (pair < total && sub == 0) {
    const long rq = pair / H;
    const int head = pair % H;
    const long b = rq / Sq, s = rq % Sq;
    delta[(b * H + head) * Sq + s] = acc;
  }
}


// Scheduling-group masks of __builtin_amdgcn_sched_group_barrier
#define SG_VALU 0x2
#define SG_MFMA 0x8
#define SG_DSR 0x100
#define SG_TRANS 0x400

// The key-owner block, hand-phased (round 1's plain-loop kernel, removed, gave bitwise the same results).  Per 32-query sub-tile
// the 16 MFMAs run as four dependent chains of 4 and every LDS fragment is read one chain (>= 128 matrix-pipe cycles) before the
// MFMA that consumes it, so no matrix instruction waits on an LDS round trip:
//   chain 1  S'  = Q K^T - lse     | reads: dO^T fragments (for chain 3)
//   chain 2  dP' = dO V^T - delta  | exp2(S') -> P, convert P         | reads: Q^T fragments (for chain 4)
//   chain 3  dV^T += dO^T P        | dS = P * dP', convert dS         | reads: next sub-tile's Q rows + row constants
//   chain 4  dK^T += Q^T dS        |                                  | reads: next sub-tile's dO rows + row constants
// The next 128-query tile is copied into the other LDS buffer by LDS-DMA during sub-tiles 0..2 (two barriers per iteration: at the
// top "everyone has left the previous tile", after sub-tile 2 "the new tile has landed"), so the last sub-tile prefetches across
// the tile seam as well and the staging costs no registers.
// ---- split sweeps (round 5) ------------------------------------------------------------------------------------------------------------------------
// A tile of the pair launch is one whole sweep: ~170 us for an encoder tile, and 1,344 of them on 512 block slots leave the chip a partial last round
// (profiles/r04_attn_pair_rounds.txt: ~10 % of the encoder launch).  The LAST tiles of each role can therefore run as two half sweeps - a key-owner tile over
// the first / second half of the query tiles, a query-owner tile over the first / second half of the key tiles - by two workgroups: the first (`part` 0)
// publishes its fp32 accumulators through scratch (register order, 1 KiB per wave-instruction, write-through stores, every wave drained, barrier, ONE flag
// word), the second (`part` 1) waits for the flag (one lane polls, agent-scope acquire, barrier), adds them to its own and stores the gradients.  Roles are
// fixed, two addends commute: run-to-run identical; against the whole sweep the association of the sum differs (fp32, before the one bf16 rounding).
// Producers precede their consumers in the tile queue, wait for nothing, and a unit is only ever claimed by a RUNNING block: no cycle.
// (SweepPart and its hand-off helpers - sweep_publish / sweep_collect - live in smbv_common.h: the forward kernel splits its last tiles the same way)

template <int QT, int DH>
struct DkvLds { static constexpr int BUF = 2 * QT * 2 * DH + 2 * QT * 4, BYTES = 2 * BUF; };

// one key-owner block: keys 128 tile_x .. + 127 of (batch b, head); smem: DkvLds<QT, DH>::BYTES, 16-byte aligned
template <int QT, int DH, bool ROPE = false>   // ROPE: dK leaves through the transposed rotary map (V-JEPA; RopeStore, smbv_common.h)
__device__ __forceinline__ void attn_dkv2_block(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                const bf16_t* __restrict__ V, int ld, const bf16_t* __restrict__ dO,
                                                int ldo, const float* __restrict__ LSE, const float* __restrict__ delta,
                                                bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, int ldd, int S, int Sq, int qoff,
                                                int H, float scale, const int tile_x, const int head, const int b, unsigned char* smem,
                                                const SweepPart sp = SweepPart(), const RopeStore rs = RopeStore()) {
  static_assert(QT == 128, "the mid-iteration hand-over is written for 4 sub-tiles");
  static_assert(DH == 64 || DH == 32, "head widths with a native build");
  constexpr int NS = QT / 32;
  constexpr int KS = DH / 16, DT = DH / 32, RB = 2 * DH;  // QK^T-type k-steps, 32-channel output tiles, LDS row bytes
  constexpr int NPC = 32 * RB / 1024;                     // LDS-DMA pieces per wave and image (32 rows)
  constexpr int BUF = DkvLds<QT, DH>::BUF;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int r = l & 31, h2 = l >> 5;
  const size_t row0 = (size_t)b * S;
  const int key = tile_x * 128 + w * 32 + r;
  const int hoff = head * DH;
  const float c = scale * 1.4426950408889634f;
  const float* lse_p = LSE + ((size_t)b * H + head) * Sq;
  const float* del_p = delta + ((size_t)b * H + head) * Sq;

  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    u32x4 a = {0, 0, 0, 0}, bb = {0, 0, 0, 0};
    if (key < S) {
      a = *(const u32x4*)(K + (row0 + key) * ld + hoff + 16 * ks + 8 * h2);
      bb = *(const u32x4*)(V + (row0 + key) * ld + hoff + 16 * ks + 8 * h2);
    }
    asm volatile("" : "+v"(bb));  // (consumed here so that hipcc waits for the load here, not with a vmcnt(0) inside the tile loop)
    kf[ks] = scale_frag(a, c);
    vf[ks] = __builtin_bit_cast(bf16x8, bb);
  }
  f32x16 dkt[DT], dvt[DT];
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) { dkt[dt][i] = 0.f; dvt[dt][i] = 0.f; }

  // Q / dO tiles go global -> LDS by LDS-DMA: wave w stages rows 32w..32w+31 of both images as 4 + 4 pieces of 8 rows x 128 B.
  // A piece lands lane-linear (lane -> row l>>3, chunk position l&7), so the swz128 permutation is applied to the SOURCE chunk.
  // Rows >= Sq are outside the descriptors' range and arrive as zeros.
  const i32x4 rsq = dma_rsrc(Q + (row0 + qoff) * ld + hoff, (unsigned)(((size_t)(Sq - 1) * ld + DH) * 2));
  const i32x4 rsd = dma_rsrc(dO + (row0 + qoff) * ldo + hoff, (unsigned)(((size_t)(Sq - 1) * ldo + DH) * 2));
  int oq[NPC], od[NPC];
#pragma unroll
  for (int i = 0; i < NPC; i++) {
    oq[i] = dma_src_off<DH>(w, i, l, ld);
    od[i] = dma_src_off<DH>(w, i, l, ldo);
  }
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem) + (t >> 6) * (32 * RB)));
  // The copies and the row-constant load are issued from asm statements (smbv_common.h): hipcc would otherwise drain them with a
  // vmcnt(0) in front of the very next LDS read.  They are waited for once, explicitly, before the "landed" barrier.
  float rraw = 0.f;
  auto stage = [&](int buf, int qt) __attribute__((always_inline)) {
    const unsigned dq_ = lds_w + (unsigned)buf * BUF, dd_ = dq_ + QT * RB;
    const int kq = qt * QT * ld * 2, kd = qt * QT * ldo * 2;
#pragma unroll
    for (int i = 0; i < NPC; i++) {
      dma16_hidden(rsq, dq_ + 1024 * i, oq[i] + kq);
      dma16_hidden(rsd, dd_ + 1024 * i, od[i] + kd);
    }
    const int qi = min(qt * QT + (t & (QT - 1)), Sq - 1);  // threads 0..QT-1: lse, QT..2QT-1: delta (rows past Sq: clamped, fixed up below)
    const float* src = (t < QT ? lse_p : del_p) + qi;
    asm volatile("global_load_dword %0, %1, off" : "=v"(rraw) : "v"(src) : "memory");
  };
  float rstat = 0.f;
  auto landed = [&](int qt) __attribute__((always_inline)) {  // everything stage(.., qt) requested has arrived (this wave's share)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rraw) : : "memory");
    const bool ok = qt * QT + (t & (QT - 1)) < Sq;
    rstat = t < QT ? (ok ? -rraw * 1.4426950408889634f : -INFINITY) : (ok ? -rraw : 0.f);  // padded rows: p = exp2(-inf) = 0
  };
  // row constants, arranged so that each lane half reads its 16 accumulator rows (8*rg + 4*h2 + e) of a 32-row sub-tile as ONE
  // contiguous 64-byte vector: position = 32*sub + 16*h2 + 4*rg + e
  auto put_stats = [&](int buf) __attribute__((always_inline)) {
    if (t < 2 * QT) {
      const int rr = t % QT;
      const int pos = (rr & ~31) + 16 * ((rr >> 2) & 1) + 4 * ((rr & 31) >> 3) + (rr & 3);
      ((float*)(smem + buf * BUF + 2 * QT * RB))[(t / QT) * QT + pos] = rstat;
    }
  };

  const int nq = sp.t1 > 0 ? sp.t1 : (Sq + QT - 1) / QT;   // this block sweeps query tiles [q_first, nq)
  const int q_first = sp.t1 > 0 ? sp.t0 : 0;
  const int g1 = (l >> 4) & 1, qq = (l >> 2) & 3, pp = l & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // transposed fragment (s2, dt) of a row-major image: rows 16*s2 + 4*h2 + qq (+8), 16-byte chunk 4*dt + 2*g1 + (pp>>1)
  auto tr_frag = [&](const unsigned char* img, int rb, int s2, int dt) __attribute__((always_inline)) -> bf16x8 {
    const int qb = rb + 16 * s2 + 4 * h2 + qq;
    const int c16 = 4 * dt + 2 * g1 + (pp >> 1);
    const int sb8 = 8 * (pp & 1);
    const s16x4 a0 = lds_read_tr16(img + swzr<DH>(qb, c16) + sb8);
    const s16x4 a1 = lds_read_tr16(img + swzr<DH>(qb + 8, c16) + sb8);
    const s16x8 aa = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    return __builtin_bit_cast(bf16x8, aa);
  };

  stage(0, q_first);
  landed(q_first);
  put_stats(0);
  __syncthreads();
  bf16x8 qa[KS], da[KS];
  f32x16 s, dp;
  {
    const unsigned char* sq = smem;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      qa[ks] = *(const bf16x8*)(sq + swzr<DH>(r, 2 * ks + h2));
      da[ks] = *(const bf16x8*)(sq + QT * RB + swzr<DH>(r, 2 * ks + h2));
    }
    s = *(const f32x16*)((const float*)(sq + 2 * QT * RB) + 16 * h2);
    dp = *(const f32x16*)((const float*)(sq + 2 * QT * RB) + QT + 16 * h2);
  }
  for (int qt = q_first; qt < nq; qt++) {
    const int buf = (qt - q_first) & 1;
    const bool more = qt + 1 < nq;
    __syncthreads();  // every wave has left the previous tile: its buffer may be overwritten
    if (more) stage(buf ^ 1, qt + 1);
    const unsigned char* sq = smem + buf * BUF;
    const unsigned char* sdo = sq + QT * RB;
#pragma unroll
    for (int sub = 0; sub < NS; sub++) {
      const int rb = 32 * sub;
      // where the NEXT sub-tile lives: same buffer, or rows 0..31 of the other buffer across the tile seam
      const unsigned char* nq_img = (sub + 1 < NS) ? sq : smem + (buf ^ 1) * BUF;
      const int nrb = (sub + 1 < NS) ? rb + 32 : 0;
      const bool pre = (sub + 1 < NS) || more;
      bf16x8 ta[2][DT], tb[2][DT], pf[2], dsf[2];
      // ---- chain 1
#pragma unroll
      for (int ks = 0; ks < KS; ks++) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[ks], s, 0, 0, 0);
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int dt = 0; dt < DT; dt++) ta[s2][dt] = tr_frag(sdo, rb, s2, dt);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
        __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- chain 2
#pragma unroll
      for (int ks = 0; ks < KS; ks++) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ks], vf[ks], dp, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; i++) s[i] = fast_exp2(s[i]);
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int j = 0; j < 8; j++) pf[s2][j] = (__bf16)s[8 * s2 + j];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int dt = 0; dt < DT; dt++) tb[s2][dt] = tr_frag(sq, rb, s2, dt);
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_DSR, 4, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 2, 0);
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 3, 0);
      __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
      __builtin_amdgcn_sched_group_barrier(SG_VALU, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 3, 0);
      __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
      __builtin_amdgcn_sched_group_barrier(SG_VALU, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 3, 0);
      __builtin_amdgcn_sched_group_barrier(SG_VALU, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- chain 3
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int dt = 0; dt < DT; dt++) dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[s2][dt], pf[s2], dvt[dt], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; i++) dp[i] *= s[i];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int j = 0; j < 8; j++) dsf[s2][j] = (__bf16)dp[8 * s2 + j];
      if (pre) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qa[ks] = *(const bf16x8*)(nq_img + swzr<DH>(nrb + r, 2 * ks + h2));
        s = *(const f32x16*)((const float*)(nq_img + 2 * QT * RB) + nrb + 16 * h2);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
        __builtin_amdgcn_sched_group_barrier(SG_VALU, 6, 0);
        __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- chain 4
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int dt = 0; dt < DT; dt++) dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tb[s2][dt], dsf[s2], dkt[dt], 0, 0, 0);
      if (pre) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) da[ks] = *(const bf16x8*)(nq_img + QT * RB + swzr<DH>(nrb + r, 2 * ks + h2));
        dp = *(const f32x16*)((const float*)(nq_img + 2 * QT * RB) + QT + nrb + 16 * h2);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
        __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (sub == 2) {
        if (more) { landed(qt + 1); put_stats(buf ^ 1); }
        __syncthreads();  // the new tile is visible before sub-tile 3 prefetches across the seam
      }
    }
  }

  if (sp.part >= 0) {   // half a sweep: the accumulators travel as 4 DT pieces of 16 bytes per lane (dK^T tiles, then dV^T tiles)
    f32x4 pv[8 * DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        pv[4 * dt + rg] = (f32x4){dkt[dt][4 * rg], dkt[dt][4 * rg + 1], dkt[dt][4 * rg + 2], dkt[dt][4 * rg + 3]};
        pv[4 * DT + 4 * dt + rg] = (f32x4){dvt[dt][4 * rg], dvt[dt][4 * rg + 1], dvt[dt][4 * rg + 2], dvt[dt][4 * rg + 3]};
      }
    __syncthreads();   // (the tile loop's LDS is dead: its first word carries the consumer's verdict)
    if (sp.part == 0) { sweep_publish<8 * DT>(sp, pv, t, w, l); return; }
    if (!sweep_collect<8 * DT>(sp, pv, t, w, l, (int*)smem)) return;
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
#pragma unroll
        for (int e = 0; e < 4; e++) { dkt[dt][4 * rg + e] = pv[4 * dt + rg][e]; dvt[dt][4 * rg + e] = pv[4 * DT + 4 * dt + rg][e]; }
  }
  if constexpr (ROPE) {   // the cos | sin table into the tile loop's (dead) LDS, behind the word the hand-off's verdict travels through
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no LDS-DMA piece of the loop is still on its way into this memory)
    __syncthreads();
    rope_stage_table((float*)(smem + 16), rs.table, rs.geo, t, 256);
    __syncthreads();
  }
  if (key < S) {
    bf16_t* kp = dK + (row0 + key) * ldd + hoff;
    bf16_t* vp = dV + (row0 + key) * ldd + hoff;
    RopeTok tk = {};
    if constexpr (ROPE) tk = rope_tok(rs.pos ? rs.pos[row0 + key] : key, rs.geo);
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int d0 = 32 * dt + 8 * rg + 4 * h2;
        f32x4 kk = {dkt[dt][4 * rg] * scale, dkt[dt][4 * rg + 1] * scale, dkt[dt][4 * rg + 2] * scale, dkt[dt][4 * rg + 3] * scale};
        // (the gradient of the ROTATED key, rounded to bf16 as the stand-alone launch read it, through the transposed map; table in LDS)
        if constexpr (ROPE) rope_round_apply4(kk, rope_lane(d0, rs.geo.seg, true), tk, rs.geo.seg, (const float*)(smem + 16), true);
        u32x2 pk = {pack_bf16(kk[0], kk[1]), pack_bf16(kk[2], kk[3])};
        *(u32x2*)(kp + d0) = pk;
        u32x2 pv = {pack_bf16(dvt[dt][4 * rg], dvt[dt][4 * rg + 1]), pack_bf16(dvt[dt][4 * rg + 2], dvt[dt][4 * rg + 3])};
        *(u32x2*)(vp + d0) = pv;
      }
  }
}


template <int QT, int DH>
__global__ __launch_bounds__(256, 2) void attn_dkv2_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                           const bf16_t* __restrict__ V, int ld, const bf16_t* __restrict__ dO,
                                                           int ldo, const float* __restrict__ LSE, const float* __restrict__ delta,
                                                           bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, int ldd, int S, int Sq, int qoff,
                                                           int H, float scale, int xcd_order) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DkvLds<QT, DH>::BYTES];
  int tile_x = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  stagger_by_wave_slot((xcd_order >> 8) & 0xff);
  if (xcd_order & 1) xcd_block_order((S + 127) / 128, H, tile_x, head, b);
  attn_dkv2_block<QT, DH>(Q, K, V, ld, dO, ldo, LSE, delta, dK, dV, ldd, S, Sq, qoff, H, scale, tile_x, head, b, smem);
}


// The query-owner block, hand-phased and rotated (round 1's plain-loop kernel, removed, gave bitwise the same results).  Steady state,
// per 32-key sub-tile j, three dependent chains of 4 MFMAs, each with the other sub-tile's VALU work and the LDS reads of a LATER
// chain in its gaps:
//   chain A  S'^T(j+1) = K Q^T - lse     | dS(j) = P(j) * dP'(j), convert          | reads: V rows of j+1
//   chain B  dQ^T += K^T dS^T(j)         | exp2 of half of S'^T(j+1)               | reads: K rows of j+2
//   chain C  dP'^T(j+1) = V dO^T - delta | exp2 of the other half                  | reads: K^T fragments of j+1
// K/V tiles of 128 keys arrive by LDS-DMA two tiles ahead into a 2-buffer ring (barriers: before body 2 "tile T+1 has landed",
// before body 3 "tile T is dead", after which tile T+2 is requested into its buffer).  The pipeline's last body runs its A and C
// chains on a stale sub-tile; their results are never used.
template <int KT, int DH>
struct DqLds { static constexpr int BUF = 2 * KT * 2 * DH, BYTES = 2 * BUF; };

// one query-owner block: queries 128 tile_x .. + 127 of the range of (batch b, head); smem: DqLds<KT, DH>::BYTES, 16-byte aligned
template <int KT, int DH>
__device__ __forceinline__ void attn_dq2_block(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                               const bf16_t* __restrict__ V, int ld, const bf16_t* __restrict__ dO, int ldo,
                                               const float* __restrict__ LSE, const float* __restrict__ delta,
                                               bf16_t* __restrict__ dQ, int ldd, int S, int Sq, int qoff, int H, float scale,
                                               const int tile_x, const int head, const int b, unsigned char* smem) {
  static_assert(KT == 128, "bodies are unrolled for 4 sub-tiles per tile");
  static_assert(DH == 64 || DH == 32, "head widths with a native build");
  constexpr int KS = DH / 16, DT = DH / 32, RB = 2 * DH;  // QK^T-type k-steps, 32-channel output tiles, LDS row bytes
  constexpr int NPC = 32 * RB / 1024;                     // LDS-DMA pieces per wave and image (32 rows)
  constexpr int BUF = DqLds<KT, DH>::BUF;  // K image | V image
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int r = l & 31, h2 = l >> 5;
  const size_t row0 = (size_t)b * S;
  const int q = tile_x * 128 + w * 32 + r;
  const int hoff = head * DH;
  const float c = scale * 1.4426950408889634f;

  bf16x8 qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    u32x4 a = {0, 0, 0, 0}, bb = {0, 0, 0, 0};
    if (q < Sq) {
      a = *(const u32x4*)(Q + (row0 + qoff + q) * ld + hoff + 16 * ks + 8 * h2);
      bb = *(const u32x4*)(dO + (row0 + qoff + q) * ldo + hoff + 16 * ks + 8 * h2);
    }
    asm volatile("" : "+v"(bb));  // (consumed here so that hipcc waits for the load here, not with a vmcnt(0) inside the tile loop)
    qf[ks] = scale_frag(a, c);
    dof[ks] = __builtin_bit_cast(bf16x8, bb);
  }
  float nlse2 = -INFINITY, ndl = 0.f;
  if (q < Sq) {
    nlse2 = -LSE[((size_t)b * H + head) * Sq + q] * 1.4426950408889634f;
    ndl = -delta[((size_t)b * H + head) * Sq + q];
  }
  asm volatile("" : "+v"(nlse2), "+v"(ndl));
  f32x16 sinit, dinit;
#pragma unroll
  for (int i = 0; i < 16; i++) { sinit[i] = nlse2; dinit[i] = ndl; }
  f32x16 dqt[DT];
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) dqt[dt][i] = 0.f;

  // LDS-DMA staging: wave w copies rows 32w..32w+31 of the K and V tiles (4 + 4 pieces of 8 rows x 128 B, swz128 applied to the
  // source chunk); rows >= S are outside the descriptors and arrive as zeros.
  const i32x4 rsk = dma_rsrc(K + row0 * ld + hoff, (unsigned)(((size_t)(S - 1) * ld + DH) * 2));
  const i32x4 rsv = dma_rsrc(V + row0 * ld + hoff, (unsigned)(((size_t)(S - 1) * ld + DH) * 2));
  int og[NPC];
#pragma unroll
  for (int i = 0; i < NPC; i++) og[i] = dma_src_off<DH>(w, i, l, ld);
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem) + (t >> 6) * (32 * RB)));
  // (issued from asm statements, see smbv_common.h: hipcc would drain a visible LDS-DMA in front of the next LDS read)
  auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
    const unsigned dk_ = lds_w + (unsigned)buf * BUF, dv_ = dk_ + KT * RB;
    const int ko = kt * KT * ld * 2;
#pragma unroll
    for (int i = 0; i < NPC; i++) {
      dma16_hidden(rsk, dk_ + 1024 * i, og[i] + ko);
      dma16_hidden(rsv, dv_ + 1024 * i, og[i] + ko);
    }
  };

  const int g1 = (l >> 4) & 1, qq = (l >> 2) & 3, pp = l & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto rows = [&](const unsigned char* img, int sub, bf16x8* f) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < KS; ks++) f[ks] = *(const bf16x8*)(img + swzr<DH>(32 * sub + r, 2 * ks + h2));
  };
  auto trk = [&](const unsigned char* img, int sub, bf16x8 (*f)[DT]) __attribute__((always_inline)) {
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const int kb = 32 * sub + 16 * s2 + 4 * h2 + qq;
        const int c16 = 4 * dt + 2 * g1 + (pp >> 1);
        const int sb8 = 8 * (pp & 1);
        const s16x4 a0 = lds_read_tr16(img + swzr<DH>(kb, c16) + sb8);
        const s16x4 a1 = lds_read_tr16(img + swzr<DH>(kb + 8, c16) + sb8);
        const s16x8 aa = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        f[s2][dt] = __builtin_bit_cast(bf16x8, aa);
      }
  };

  const int nt = (S + KT - 1) / KT;
  stage(0, 0);
  if (nt > 1) stage(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 ka[KS], va[KS], tk[2][DT];
  f32x16 sA, sB, dp;
  {
    rows(smem, 0, ka);
    rows(smem + KT * RB, 0, va);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks], ks == 0 ? sinit : sA, 0, 0, 0);
    rows(smem, 1, ka);
    trk(smem, 0, tk);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[ks], dof[ks], ks == 0 ? dinit : dp, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++) sA[i] = fast_exp2(sA[i]);
  }
  // one pipeline body: finishes sub-tile j (P in sC, dP' in dp, K^T fragments in tk) and starts j+1 (K rows in ka)
  auto body = [&](f32x16& sC, f32x16& sN, const unsigned char* img1, int sub1, const unsigned char* img2, int sub2) __attribute__((always_inline)) {
    bf16x8 dsf[2];
    // ---- chain A
#pragma unroll
    for (int ks = 0; ks < KS; ks++) sN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks], ks == 0 ? sinit : sN, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++) dp[i] *= sC[i];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int j = 0; j < 8; j++) dsf[s2][j] = (__bf16)dp[8 * s2 + j];
    rows(img1 + KT * RB, sub1, va);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_VALU, 6, 0);
      __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- chain B
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int dt = 0; dt < DT; dt++) dqt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tk[s2][dt], dsf[s2], dqt[dt], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; i++) { sN[i] = fast_exp2(sN[i]); asm volatile("" : "+v"(sN[i])); }  // (pinned: hipcc otherwise sinks the last body's exps into the next iteration's head)
    rows(img2, sub2, ka);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 2, 0);
      __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- chain C
#pragma unroll
    for (int ks = 0; ks < KS; ks++) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[ks], dof[ks], ks == 0 ? dinit : dp, 0, 0, 0);
#pragma unroll
    for (int i = 8; i < 16; i++) { sN[i] = fast_exp2(sN[i]); asm volatile("" : "+v"(sN[i])); }
    trk(img1, sub1, tk);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 2, 0);
      __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // one 128-key tile out of buffer `cb` (a literal at both call sites: every LDS read is a per-lane base plus an immediate)
  auto tile = [&](int kt, int cb) __attribute__((always_inline)) {
    const unsigned char* cur = smem + cb * BUF;
    const unsigned char* nxt = smem + (cb ^ 1) * BUF;
    body(sA, sB, cur, 1, cur, 2);
    body(sB, sA, cur, 2, cur, 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt+1 (requested a whole tile ago) have landed
    __syncthreads();                                  // ... everyone's have
    body(sA, sB, cur, 3, nxt, 0);
    __syncthreads();  // every wave is done with tile kt: its buffer takes tile kt+2
    if (kt + 2 < nt) stage(cb, kt + 2);
    body(sB, sA, nxt, 0, nxt, 1);
  };
  for (int kt = 0; kt < nt; kt += 2) {
    tile(kt, 0);
    if (kt + 1 < nt) tile(kt + 1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup

  if (q < Sq) {
    bf16_t* qp = dQ + (row0 + qoff + q) * ldd + hoff;
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int d0 = 32 * dt + 8 * rg + 4 * h2;
        u32x2 pk = {pack_bf16(dqt[dt][4 * rg] * scale, dqt[dt][4 * rg + 1] * scale), pack_bf16(dqt[dt][4 * rg + 2] * scale, dqt[dt][4 * rg + 3] * scale)};
        *(u32x2*)(qp + d0) = pk;
      }
  }
}

// The query-owner block on v_mfma_f32_16x16x32_bf16 (round 3, the shipped one): a lane is (query n = l & 15 of a 16-query tile, contraction group
// g = l >> 4); an S'^T / dP'^T tile (key tile kj, query tile qi) holds keys 16 kj + 4 g + e, so the two key tiles' dS^T accumulators of one query tile,
// converted, are the B operand (contraction index 8 g + j <-> key 4 g + j | 16 + 4 g + j - 4) of dQ^T += K^T dS^T, whose A operand is two
// ds_read_b64_tr_b16 blocks of K rows 4 g .. and 16 + 4 g ...  Same rotated three-chain pipeline, K / V ring and products as attn_dq2_block (the
// 32x32x16 form, debug bit 2), twice the MFMA instructions, 203 instead of 226 registers.  Why: the chip holds a higher clock on the 16x16x32 shape
// (an accumulator register is written once per 64 FLOP instead of once per 32; tools/probe/mfma_shape_probe.hip), at more cycles.  Measured in the
// step, where the attention launches follow low-power kernels and cycles count for more than in a back-to-back loop: -0.1 ms MIM, -0.9 ms V-JEPA
// for this block; the key-owner block re-tiled the same way was 4 % faster back to back on random operands and 0.15-0.3 ms SLOWER in the step, and
// stays on 32x32x16 (profiles/r03_mfma_shape.txt).
template <int KT, int DH, bool ROPE = false>   // ROPE: dQ leaves through the transposed rotary map
__device__ __forceinline__ void attn_dq3_block(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                               const bf16_t* __restrict__ V, int ld, const bf16_t* __restrict__ dO, int ldo,
                                               const float* __restrict__ LSE, const float* __restrict__ delta,
                                               bf16_t* __restrict__ dQ, int ldd, int S, int Sq, int qoff, int H, float scale,
                                               const int tile_x, const int head, const int b, unsigned char* smem, const SweepPart sp = SweepPart(),
                                               const RopeStore rs = RopeStore()) {
  static_assert(KT == 128, "bodies are unrolled for 4 sub-tiles per tile");
  static_assert(DH == 64 || DH == 32, "head widths with a native build");
  constexpr int KK = DH / 32, DTN = DH / 16, RB = 2 * DH;
  constexpr int NPC = 32 * RB / 1024;
  constexpr int BUF = DqLds<KT, DH>::BUF;  // K image | V image
  constexpr int N1 = 4 * KK, N3 = 2 * DTN;  // MFMAs per chain: S'^T / dP'^T and dQ^T
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int n = l & 15, g = l >> 4;
  const size_t row0 = (size_t)b * S;
  const int hoff = head * DH;
  const float c = scale * 1.4426950408889634f;

  // Q (pre-scaled) / dO operands: [query tile][contraction step]; lane: query 16 qi + n of the wave's 32, channels 32 kk + 8 g .. + 7
  bf16x8 qf[2][KK], dof[2][KK];
  f32x4 sinit[2], dinit[2];
#pragma unroll
  for (int qi = 0; qi < 2; qi++) {
    const int q = tile_x * 128 + w * 32 + 16 * qi + n;
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
      u32x4 a = {0, 0, 0, 0}, bb = {0, 0, 0, 0};
      if (q < Sq) {
        a = *(const u32x4*)(Q + (row0 + qoff + q) * ld + hoff + 32 * kk + 8 * g);
        bb = *(const u32x4*)(dO + (row0 + qoff + q) * ldo + hoff + 32 * kk + 8 * g);
      }
      asm volatile("" : "+v"(bb));  // (consumed here so that hipcc waits for the load here, not with a vmcnt(0) inside the tile loop)
      qf[qi][kk] = scale_frag(a, c);
      dof[qi][kk] = __builtin_bit_cast(bf16x8, bb);
    }
    float nlse2 = -INFINITY, ndl = 0.f;
    if (q < Sq) {
      nlse2 = -LSE[((size_t)b * H + head) * Sq + q] * 1.4426950408889634f;
      ndl = -delta[((size_t)b * H + head) * Sq + q];
    }
    asm volatile("" : "+v"(nlse2), "+v"(ndl));
    sinit[qi] = (f32x4){nlse2, nlse2, nlse2, nlse2};
    dinit[qi] = (f32x4){ndl, ndl, ndl, ndl};
  }
  f32x4 dqt[DTN][2];
#pragma unroll
  for (int dt = 0; dt < DTN; dt++)
#pragma unroll
    for (int qi = 0; qi < 2; qi++) dqt[dt][qi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // half a sweep (sp.t1 > 0): key tiles [sp.t0, sp.t1) - the descriptors start at the first of them and end with the last (rows past it read zeros: no contribution)
  const int k_first = sp.t1 > 0 ? sp.t0 * KT : 0;
  const int Sk = (sp.t1 > 0 ? min(S, sp.t1 * KT) : S) - k_first;
  const i32x4 rsk = dma_rsrc(K + (row0 + k_first) * ld + hoff, (unsigned)(((size_t)(Sk - 1) * ld + DH) * 2));
  const i32x4 rsv = dma_rsrc(V + (row0 + k_first) * ld + hoff, (unsigned)(((size_t)(Sk - 1) * ld + DH) * 2));
  int og[NPC];
#pragma unroll
  for (int i = 0; i < NPC; i++) og[i] = dma_src_off<DH>(w, i, l, ld);
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem) + (t >> 6) * (32 * RB)));
  auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
    const unsigned dk_ = lds_w + (unsigned)buf * BUF, dv_ = dk_ + KT * RB;
    const int ko = kt * KT * ld * 2;
#pragma unroll
    for (int i = 0; i < NPC; i++) {
      dma16_hidden(rsk, dk_ + 1024 * i, og[i] + ko);
      dma16_hidden(rsv, dv_ + 1024 * i, og[i] + ko);
    }
  };

  const int qq = (l >> 2) & 3, pp = l & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // row fragments of sub-tile `sub`: [key tile kj][contraction step kk]; lane: key 16 kj + n, channels 32 kk + 8 g .. + 7
  auto rows = [&](const unsigned char* img, int sub, bf16x8 (*f)[KK]) __attribute__((always_inline)) {
#pragma unroll
    for (int kj = 0; kj < 2; kj++)
#pragma unroll
      for (int kk = 0; kk < KK; kk++) f[kj][kk] = *(const bf16x8*)(img + swzr<DH>(32 * sub + 16 * kj + n, 4 * kk + g));
  };
  // transposed K fragments: channels 16 dt + n, contraction over keys {4 g .. + 3} and {16 + 4 g .. + 3} of the sub-tile
  auto trk = [&](const unsigned char* img, int sub, bf16x8* f) __attribute__((always_inline)) {
#pragma unroll
    for (int dt = 0; dt < DTN; dt++) {
      const int kb = 32 * sub + 4 * g + qq;
      const int c16 = 2 * dt + (pp >> 1);
      const int sb8 = 8 * (pp & 1);
      const s16x4 a0 = lds_read_tr16(img + swzr<DH>(kb, c16) + sb8);
      const s16x4 a1 = lds_read_tr16(img + swzr<DH>(kb + 16, c16) + sb8);
      const s16x8 aa = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      f[dt] = __builtin_bit_cast(bf16x8, aa);
    }
  };
  struct Tiles { f32x4 v[2][2]; };  // [key tile kj][query tile qi]
  auto chain_s = [&](Tiles& o, const bf16x8 (*a)[KK], const bf16x8 (*bq)[KK], const f32x4* init) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < KK; kk++)
#pragma unroll
      for (int kj = 0; kj < 2; kj++)
#pragma unroll
        for (int qi = 0; qi < 2; qi++)
          o.v[kj][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kj][kk], bq[qi][kk], kk == 0 ? init[qi] : o.v[kj][qi], 0, 0, 0);
  };

  const int nt = (Sk + KT - 1) / KT;
  stage(0, 0);
  if (nt > 1) stage(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 ka[2][KK], va[2][KK], tk[DTN];
  Tiles sA, sB, dp;
  {
    rows(smem, 0, ka);
    rows(smem + KT * RB, 0, va);
    chain_s(sA, ka, qf, sinit);
    rows(smem, 1, ka);
    trk(smem, 0, tk);
    chain_s(dp, va, dof, dinit);
#pragma unroll
    for (int kj = 0; kj < 2; kj++)
#pragma unroll
      for (int qi = 0; qi < 2; qi++)
#pragma unroll
        for (int e = 0; e < 4; e++) sA.v[kj][qi][e] = fast_exp2(sA.v[kj][qi][e]);
  }
  // one pipeline body: finishes sub-tile j (P in sC, dP' in dp, K^T fragments in tk) and starts j+1 (K rows in ka)
  auto body = [&](Tiles& sC, Tiles& sN, const unsigned char* img1, int sub1, const unsigned char* img2, int sub2) __attribute__((always_inline)) {
    bf16x8 dsf[2];
    // ---- chain A: S'^T(j+1) | dS(j) = P(j) * dP'(j), convert | reads: V rows of j+1
    chain_s(sN, ka, qf, sinit);
#pragma unroll
    for (int qi = 0; qi < 2; qi++) {
#pragma unroll
      for (int kj = 0; kj < 2; kj++)
#pragma unroll
        for (int e = 0; e < 4; e++) dp.v[kj][qi][e] *= sC.v[kj][qi][e];
#pragma unroll
      for (int e = 0; e < 4; e++) { dsf[qi][e] = (__bf16)dp.v[0][qi][e]; dsf[qi][4 + e] = (__bf16)dp.v[1][qi][e]; }
    }
    rows(img1 + KT * RB, sub1, va);
#pragma unroll
    for (int i = 0; i < N1; i++) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_VALU, 24 / N1, 0);
      if (i < 2 * KK) __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- chain B: dQ^T += K^T dS^T(j) | exp2 of the first key tile of S'^T(j+1) | reads: K rows of j+2
#pragma unroll
    for (int qi = 0; qi < 2; qi++)
#pragma unroll
      for (int dt = 0; dt < DTN; dt++) dqt[dt][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tk[dt], dsf[qi], dqt[dt][qi], 0, 0, 0);
#pragma unroll
    for (int qi = 0; qi < 2; qi++)
#pragma unroll
      for (int e = 0; e < 4; e++) { sN.v[0][qi][e] = fast_exp2(sN.v[0][qi][e]); asm volatile("" : "+v"(sN.v[0][qi][e])); }  // (pinned: hipcc otherwise sinks the last body's exps into the next iteration's head)
    rows(img2, sub2, ka);
#pragma unroll
    for (int i = 0; i < N3; i++) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 8 / N3, 0);
      if (i < 2 * KK) __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- chain C: dP'^T(j+1) | exp2 of the second key tile | reads: K^T fragments of j+1
    chain_s(dp, va, dof, dinit);
#pragma unroll
    for (int qi = 0; qi < 2; qi++)
#pragma unroll
      for (int e = 0; e < 4; e++) { sN.v[1][qi][e] = fast_exp2(sN.v[1][qi][e]); asm volatile("" : "+v"(sN.v[1][qi][e])); }
    trk(img1, sub1, tk);
#pragma unroll
    for (int i = 0; i < N1; i++) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 8 / N1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_DSR, 2 * DTN / N1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tile = [&](int kt, int cb) __attribute__((always_inline)) {
    const unsigned char* cur = smem + cb * BUF;
    const unsigned char* nxt = smem + (cb ^ 1) * BUF;
    body(sA, sB, cur, 1, cur, 2);
    body(sB, sA, cur, 2, cur, 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt+1 (requested a whole tile ago) have landed
    __syncthreads();                                  // ... everyone's have
    body(sA, sB, cur, 3, nxt, 0);
    __syncthreads();  // every wave is done with tile kt: its buffer takes tile kt+2
    if (kt + 2 < nt) stage(cb, kt + 2);
    body(sB, sA, nxt, 0, nxt, 1);
  };
  for (int kt = 0; kt < nt; kt += 2) {
    tile(kt, 0);
    if (kt + 1 < nt) tile(kt + 1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
  if (sp.part >= 0) {   // half a sweep: dQ^T as 2 DTN pieces of 16 bytes per lane
    f32x4 pv[2 * DTN];
#pragma unroll
    for (int dt = 0; dt < DTN; dt++)
#pragma unroll
      for (int qi = 0; qi < 2; qi++) pv[2 * dt + qi] = dqt[dt][qi];
    __syncthreads();
    if (sp.part == 0) { sweep_publish<2 * DTN>(sp, pv, t, w, l); return; }
    if (!sweep_collect<2 * DTN>(sp, pv, t, w, l, (int*)smem)) return;
#pragma unroll
    for (int dt = 0; dt < DTN; dt++)
#pragma unroll
      for (int qi = 0; qi < 2; qi++) dqt[dt][qi] = pv[2 * dt + qi];
  }

  if constexpr (ROPE) {   // the cos | sin table into the tile loop's (dead) LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no LDS-DMA piece of the loop is still on its way into this memory)
    __syncthreads();
    rope_stage_table((float*)(smem + 16), rs.table, rs.geo, t, 256);
    __syncthreads();
  }
#pragma unroll
  for (int qi = 0; qi < 2; qi++) {
    const int q = tile_x * 128 + w * 32 + 16 * qi + n;
    if (q < Sq) {
      bf16_t* qp = dQ + (row0 + qoff + q) * ldd + hoff;
      RopeTok tk = {};
      if constexpr (ROPE) tk = rope_tok(rs.pos ? rs.pos[row0 + qoff + q] : qoff + q, rs.geo);
#pragma unroll
      for (int dt = 0; dt < DTN; dt++) {
        const int d0 = 16 * dt + 4 * g;
        f32x4 qq = dqt[dt][qi] * scale;
        if constexpr (ROPE) rope_round_apply4(qq, rope_lane(d0, rs.geo.seg, true), tk, rs.geo.seg, (const float*)(smem + 16), true);
        u32x2 pk = {pack_bf16(qq[0], qq[1]), pack_bf16(qq[2], qq[3])};
        *(u32x2*)(qp + d0) = pk;
      }
    }
  }
}

// The query-owner block with its two RECOMPUTED products on the block-scaled fp8 MFMA (round 4; BASELINE.json config 5, "fp8 MFMA QK^T / PV" - in the
// backward: S'^T = K Q^T and dP'^T = V dO^T, each ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32 keys x 32 queries instead of four bf16 MFMAs, at twice the rate).
// Operands: e4m3 rows of Q (pre-scaled into the exp2 domain), K, V, dO with one E8M0 scale per (token, head, 32 channels) from attention_bwd_fp8_prepare
// (attention_fp8.hip; operand / scale-block maps as attn_fwd_fp8_kernel documents them).  The accumulator input carries -LSE log2 e / -delta, the
// accumulators come out key-on-the-row, query-on-the-lane exactly as the bf16 32x32x16 form's do, so dS^T = P (dP' ) converts straight into the B operand of
// dQ^T += K^T dS^T - which stays a bf16 product on the bf16 K image (gradients keep bf16 operand precision where they are accumulated).
// Head_dim 64.  64-key tiles through a ring of three slots: K8 [64][64 B] | V8 [64][64 B] | scales | K bf16 [64][128 B]; two tiles in flight, one counted
// vmcnt + one barrier per tile; compiler-scheduled.
struct DqF8Lds { static constexpr int SLOT = 4096 + 4096 + 256 + 8192, BYTES = 3 * SLOT; };
__device__ __forceinline__ int fp8_swz_b(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
typedef int v8i_b __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void attn_dq_fp8_block(const Fp8BwdOps f8, const bf16_t* __restrict__ K, int ld, const float* __restrict__ LSE,
                                                  const float* __restrict__ delta, bf16_t* __restrict__ dQ, int ldd, int S, int Sq, int qoff, int H, float scale,
                                                  const int tile_x, const int head, const int b, unsigned char* smem, const SweepPart sp = SweepPart()) {
  constexpr int SLOT = DqF8Lds::SLOT;
  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = l & 31, h = l >> 5;
  const int w = f8.w;
  const int q = tile_x * 128 + wv * 32 + r;
  const bool qok = q < Sq;
  const size_t qrow = (size_t)b * S + qoff + (qok ? q : 0);
  const unsigned char* base = f8.base;
  const int nsc = w >> 5;
  v8i_b qf = {0, 0, 0, 0, 0, 0, 0, 0}, dof = {0, 0, 0, 0, 0, 0, 0, 0};
  int qs = 127, dos = 127;
  float nlse2 = -INFINITY, ndl = 0.f;
  if (qok) {
    const u32x4 a = *(const u32x4*)(base + qrow * w + head * 64 + 16 * h);
    const u32x4 c = *(const u32x4*)(base + qrow * w + head * 64 + 32 + 16 * h);
    qf = (v8i_b){(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)c[0], (int)c[1], (int)c[2], (int)c[3]};
    qs = (base + f8.offSQ)[qrow * nsc + head * 2 + h];
    const u32x4 e = *(const u32x4*)(base + f8.offD + qrow * w + head * 64 + 16 * h);
    const u32x4 f = *(const u32x4*)(base + f8.offD + qrow * w + head * 64 + 32 + 16 * h);
    dof = (v8i_b){(int)e[0], (int)e[1], (int)e[2], (int)e[3], (int)f[0], (int)f[1], (int)f[2], (int)f[3]};
    dos = (base + f8.offSD)[qrow * nsc + head * 2 + h];
    nlse2 = -LSE[((size_t)b * H + head) * Sq + q] * 1.4426950408889634f;
    ndl = -delta[((size_t)b * H + head) * Sq + q];
  }
  asm volatile("" : "+v"(qf), "+v"(dof), "+v"(qs), "+v"(dos), "+v"(nlse2), "+v"(ndl));  // consumed here: no vmcnt(0) for these loads inside the tile loop
  f32x16 sinit, dinit, dqt[2];
#pragma unroll
  for (int i = 0; i < 16; i++) { sinit[i] = nlse2; dinit[i] = ndl; dqt[0][i] = 0.f; dqt[1][i] = 0.f; }

  // copy side (as attn_fwd_fp8_kernel): wave wv stages rows 16 wv .. + 15 of the K8 / V8 tiles (one 1-KiB piece each), their 2 x 32 scale bytes (lanes 0-7 K,
  // 8-15 V) and rows 16 wv .. + 15 of the bf16 K tile (two pieces of 8 rows; rows >= S arrive as zeros through the descriptor)
  const i32x4 rs = dma_rsrc(base, f8.bytes);
  const i32x4 rsk = dma_rsrc(K + (size_t)b * S * ld + head * 64, (unsigned)(((size_t)(S - 1) * ld + 64) * 2));
  const int prow = 16 * wv + (l >> 2), csrc = (l & 3) ^ ((l >> 4) & 3);
  const int ok0 = (int)f8.offK + (int)(((size_t)b * S + prow) * w) + head * 64 + 16 * csrc;
  const int ov0 = (int)f8.offV + (int)(((size_t)b * S + prow) * w) + head * 64 + 16 * csrc;
  const int j8 = l & 7;
  const int os0 = (l < 8 ? (int)f8.offSK : (int)f8.offSV) + (int)((((size_t)b * H + head) * S + 16 * wv) * 2) + 4 * j8;
  int okb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = 16 * wv + 8 * i + (l >> 3);
    const int x = (row >> 1) & 7;
    okb[i] = (row * ld + ((l & 7) ^ (((x & 1) << 2) | (x >> 1))) * 8) * 2;
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((int)lds_addr_of(smem));
  auto stage = [&](int kt, int slot) __attribute__((always_inline)) {   // slot = kt % 3, a literal at every call site
    const unsigned sl = lds0 + (unsigned)slot * SLOT;
    dma16_hidden(rs, sl + (unsigned)wv * 1024, ok0 + kt * 64 * w);
    dma16_hidden(rs, sl + 4096 + (unsigned)wv * 1024, ov0 + kt * 64 * w);
    if (l < 16) dma4_hidden(rs, sl + 8192 + (unsigned)wv * 64, os0 + kt * 128);
    dma16_hidden(rsk, sl + 8448 + (unsigned)wv * 2048, okb[0] + kt * 64 * ld * 2);
    dma16_hidden(rsk, sl + 8448 + (unsigned)wv * 2048 + 1024, okb[1] + kt * 64 * ld * 2);
  };
  const int g1 = (l >> 4) & 1, qq = (l >> 2) & 3, pp = l & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int nt = (S + 63) >> 6;                                    // 64-key tiles of the sequence
  const int kt_first = sp.t1 > 0 ? 2 * sp.t0 : 0;                  // half a sweep (sp counts 128-key tiles): this block's 64-key tiles [kt_first, kt_end)
  const int kt_end = sp.t1 > 0 ? min(nt, 2 * sp.t1) : nt;
  // One 64-key tile out of ring slot `slot` (a literal: the loop below is unrolled over the ring, so every LDS read is a per-lane base hoisted out of the
  // loop plus an immediate - with the slot computed from kt the loop carried ~45 address adds per tile next to its 80 exponentials / multiplies / converts,
  // in a block that is vector-issue bound)
  // `tail`: a literal at every call site.  Only the sequence's LAST tile can hold keys past its end, and only when S is not a multiple of 64: that tile runs
  // as a copy of its own behind the loop.  (Round 5: the test `kt == nt - 1 && (S & 63)` inside the one body had been if-converted - 32 compares and 32
  // selects per tile, a quarter of the vector instructions of a block that is vector-issue bound, for a mask that 319 of 320 tiles cannot need.)
  auto tile_at = [&](int kt, const unsigned char* sb, const bool tail) __attribute__((always_inline)) {
    const unsigned char* scb = sb + 8192;
    const unsigned char* kimg = sb + 8448;
    // both sub-tiles' fp8 products first: the exponentials / multiplies / converts of sub-tile 0 then run under sub-tile 1's MFMAs, and sub-tile 0's dQ
    // products under sub-tile 1's vector work
    f32x16 sx[2], dp[2];
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      const int row = 32 * tt + r;
      const u32x4 a = *(const u32x4*)(sb + fp8_swz_b(row, h));
      const u32x4 c = *(const u32x4*)(sb + fp8_swz_b(row, 2 + h));
      const v8i_b kf = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)c[0], (int)c[1], (int)c[2], (int)c[3]};
      const int ks = scb[(row >> 4) * 64 + (row & 15) * 2 + h];
      const u32x4 e = *(const u32x4*)(sb + 4096 + fp8_swz_b(row, h));
      const u32x4 f = *(const u32x4*)(sb + 4096 + fp8_swz_b(row, 2 + h));
      const v8i_b vf = {(int)e[0], (int)e[1], (int)e[2], (int)e[3], (int)f[0], (int)f[1], (int)f[2], (int)f[3]};
      const int vs = scb[(row >> 4) * 64 + 32 + (row & 15) * 2 + h];
      sx[tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf, sinit, 0, 0, 0, ks, 0, qs);     // S'^T[key][q] - lse, exp2 domain
      dp[tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, dof, dinit, 0, 0, 0, vs, 0, dos);   // dP'^T[key][q] - delta
    }
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
#pragma unroll
      for (int i = 0; i < 16; i++) dp[tt][i] *= fast_exp2(sx[tt][i]);
      if (tail) {  // keys past the end of the sequence: their e4m3 rows are whatever follows in the workspace
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int key = kt * 64 + 32 * tt + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= S) dp[tt][i] = 0.f;
        }
      }
      bf16x8 dsf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int j = 0; j < 8; j++) dsf[s2][j] = (__bf16)dp[tt][8 * s2 + j];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int dt = 0; dt < 2; dt++) {
          const int kb = 32 * tt + 16 * s2 + 4 * h + qq;
          const int c16 = 4 * dt + 2 * g1 + (pp >> 1);
          const int sb8 = 8 * (pp & 1);
          const s16x4 a0 = lds_read_tr16(kimg + swz128(kb, c16) + sb8);
          const s16x4 a1 = lds_read_tr16(kimg + swz128(kb + 8, c16) + sb8);
          const s16x8 aa = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          dqt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aa), dsf[s2], dqt[dt], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile kt have returned before it can reach the next barrier
  };
  auto tile = [&](int kt, auto slotc) __attribute__((always_inline)) { tile_at(kt, smem + decltype(slotc)::value * SLOT, false); };
  // this wave's five copies of tile kt have landed (tile kt + 1's may still fly); the barrier publishes the tile and retires everybody's reads of tile
  // kt - 1, whose slot tile kt + 2 now takes (past the end the K8 / V8 / scale copies read unused bytes of the workspace, the bf16 copies zeros: never used)
  auto landed = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  const bool masked_last = (S & 63) && kt_end == nt;          // this block's range ends with the sequence's ragged last tile
  const int kt_stop = masked_last ? kt_end - 1 : kt_end;      // tiles [kt_first, kt_stop): every key inside the sequence
  stage(kt_first, 0);
  stage(kt_first + 1, 1);
  for (int kt = kt_first; kt < kt_stop; kt += 3) {
    landed();
    stage(kt + 2, 2);
    tile(kt, std::integral_constant<int, 0>{});
    if (kt + 1 >= kt_stop) break;
    landed();
    stage(kt + 3, 0);
    tile(kt + 1, std::integral_constant<int, 1>{});
    if (kt + 2 >= kt_stop) break;
    landed();
    stage(kt + 4, 1);
    tile(kt + 2, std::integral_constant<int, 2>{});
  }
  if (masked_last) {   // (its copies and tile kt_stop + 1's were issued by the iterations above or the prologue: the same "five may still fly" count)
    landed();
    tile_at(kt_stop, smem + ((kt_stop - kt_first) % 3) * SLOT, true);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
  if (sp.part >= 0) {   // half a sweep: dQ^T as 8 pieces of 16 bytes per lane (the layout of the 16x16x32 block's hand-off: same scratch size)
    f32x4 pv[8];
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) pv[4 * dt + rg] = (f32x4){dqt[dt][4 * rg], dqt[dt][4 * rg + 1], dqt[dt][4 * rg + 2], dqt[dt][4 * rg + 3]};
    __syncthreads();
    if (sp.part == 0) { sweep_publish<8>(sp, pv, t, wv, l); return; }
    if (!sweep_collect<8>(sp, pv, t, wv, l, (int*)smem)) return;
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
#pragma unroll
        for (int e = 0; e < 4; e++) dqt[dt][4 * rg + e] = pv[4 * dt + rg][e];
  }
  if (qok) {
    bf16_t* qp = dQ + ((size_t)b * S + qoff + q) * ldd + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int d0 = 32 * dt + 8 * rg + 4 * h;
        u32x2 pk = {pack_bf16(dqt[dt][4 * rg] * scale, dqt[dt][4 * rg + 1] * scale), pack_bf16(dqt[dt][4 * rg + 2] * scale, dqt[dt][4 * rg + 3] * scale)};
        *(u32x2*)(qp + d0) = pk;
      }
  }
}

template <int KT, int DH, bool MFMA16>
__global__ __launch_bounds__(256, 2) void attn_dq2_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ V, int ld, const bf16_t* __restrict__ dO, int ldo,
                                                          const float* __restrict__ LSE, const float* __restrict__ delta,
                                                          bf16_t* __restrict__ dQ, int ldd, int S, int Sq, int qoff, int H, float scale, int xcd_order) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DqLds<KT, DH>::BYTES];
  int tile_x = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  stagger_by_wave_slot((xcd_order >> 8) & 0xff);
  if (xcd_order & 1) xcd_block_order((Sq + 127) / 128, H, tile_x, head, b);
  if constexpr (MFMA16) attn_dq3_block<KT, DH>(Q, K, V, ld, dO, ldo, LSE, delta, dQ, ldd, S, Sq, qoff, H, scale, tile_x, head, b, smem);
  else attn_dq2_block<KT, DH>(Q, K, V, ld, dO, ldo, LSE, delta, dQ, ldd, S, Sq, qoff, H, scale, tile_x, head, b, smem);
}

// ---- the pair as ONE launch with the two kinds of blocks mixed on every CU --------------------------------------------------------------------------
// Launched separately (dK/dV on the caller's stream, dQ forked beside it), the two grids do not really run side by side: the first kernel's blocks
// take both slots of every CU, the second kernel's blocks enter as those drain, and a CU mostly holds two blocks of the SAME kernel - two key-owner
// blocks each asking the LDS port for 96 cycles per wave and sub-tile.  Here a block learns its CU (HW_ID / XCC_ID), draws a per-CU ticket and takes
// the role the ticket's parity names - the CU's two resident blocks are then one of each kind whenever both kinds have tiles left - and claims the next
// tile of that role from a device-side queue (or of the other role once its own is exhausted).  Which block computes which tile does not change a single
// bit of the result: every tile's arithmetic is self-contained.
static constexpr int PAIR_SPLIT_MAX = 256;   // at most this many tiles of each role run as two half sweeps
struct PairState {            // one slot of the library's device-side scheduling state; a slot belongs to ONE stream of its device (attention_bwd, below)
  unsigned head[2];           // next unclaimed tile: [0] key-owner tiles, [1] query-owner tiles   } each word = (launch epoch & 0xfff) << 20 | count:
  unsigned done;              // blocks of the running launch that have ended                      } a launch only accepts words stamped with ITS epoch
  unsigned pad[13];
  unsigned cu_ticket[2048];   // arrival counter per CU, indexed by (XCC id, SE id, SH id, CU id): 3 + 3 + 1 + 4 bits (parity only: roles, never results)
  unsigned split_flag[2][PAIR_SPLIT_MAX];   // hand-off flags of the split sweeps (SweepPart): zero at module load, raised to (launch epoch + 1) by a producer, cleared again
                                            // by its consumer.  A launch that faulted or ran short (0x5001 / 0x5002 blocks run no tile, a 0x5004 consumer gives up) leaves
                                            // words behind: they carry THAT launch's epoch, and a later consumer only accepts its own
};
__device__ PairState g_pair_state[8];
static constexpr unsigned PAIR_COUNT_MASK = 0xfffffu;
// Scheduling faults are reported, not absorbed: a word that does not carry the launch's epoch (two launches in one slot at once, a slot left dirty), a
// block that finds both tile queues exhausted (some tile was skipped) - each stores a code into the device's host-mapped status word
// (streams.hip device_status_word), which every later entry point of the library returns as SMBV_ERR_STATE.  Codes: include/smbv.h smbv_device_status.
__device__ __forceinline__ void pair_fault(unsigned* status, unsigned code) {
  if (status) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ROPE (QKIND 1 only): dQ and dK are the gradients of the rotated q / k and leave through the transposed rotary map
template <int DH, int QKIND, bool ROPE = false>   // query-owner block: 0 = 32x32x16 bf16, 1 = 16x16x32 bf16 (shipped), 2 = S / dP recomputation on the fp8 MFMA (head_dim 64)
__global__ __launch_bounds__(256, 2) void attn_bwd_pair_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, int ld,
                                                               const bf16_t* __restrict__ dO, int ldo, const float* __restrict__ LSE,
                                                               const float* __restrict__ delta, bf16_t* __restrict__ dQ, bf16_t* __restrict__ dK,
                                                               bf16_t* __restrict__ dV, int ldd, int B, int S, int Sq, int qoff, int H, float scale, int slot,
                                                               unsigned epoch, unsigned* __restrict__ status, Fp8BwdOps f8, int xk, int xq, f32x4* __restrict__ split_ws,
                                                               RopeStore rs) {
  constexpr int LDS = DkvLds<128, DH>::BYTES > DqLds<128, DH>::BYTES ? DkvLds<128, DH>::BYTES : DqLds<128, DH>::BYTES;
  static_assert(DqF8Lds::BYTES <= DkvLds<128, 64>::BYTES, "the fp8 query-owner block's ring fits the key-owner block's LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS];
  PairState& ps = g_pair_state[slot];
  // units of work per role: every tile once, except the LAST xk key-owner / xq query-owner tiles, which are two half sweeps each (SweepPart): the xk (xq)
  // producer halves first, then the consumer halves - the queue's claim order
  const unsigned t_kv = (unsigned)(((S + 127) / 128) * H * B), t_q = (unsigned)(((Sq + 127) / 128) * H * B);
  const unsigned n_kv = t_kv + (unsigned)xk, n_q = t_q + (unsigned)xq;
  int* bc = (int*)smem;  // (the decision travels through LDS before any tile data does)
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15;  // HW_REG_XCC_ID
    const unsigned cu = ((xcc & 7) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
    // arrivals on a CU alternate (balancing what is RESIDENT on the CU instead - a counter every block undoes when it ends - measured the same:
    // profiles/r03_attn_pair_ab.txt)
    unsigned role = atomicAdd(&ps.cu_ticket[cu], 1u) & 1u;
    unsigned t = atomicAdd(&ps.head[role], 1u);
    bool stale = (t >> 20) != epoch;
    unsigned tile = t & PAIR_COUNT_MASK;
    if (!stale && tile >= (role ? n_q : n_kv)) {
      role ^= 1u;
      t = atomicAdd(&ps.head[role], 1u);
      stale = (t >> 20) != epoch;
      tile = t & PAIR_COUNT_MASK;
    }
    // The grid has exactly one block per tile, so with a clean slot the second queue cannot be exhausted as well.  A block that finds it so - or a word
    // of another epoch - runs no tile (never one twice, never one out of range) and raises the fault.
    if (stale) { pair_fault(status, 0x5001u); role = 2u; }
    else if (tile >= (role ? n_q : n_kv)) { pair_fault(status, 0x5002u); role = 2u; }
    bc[0] = (int)role;
    bc[1] = (int)tile;
  }
  __syncthreads();
  const int role = bc[0];
  int tile = bc[1];
  __syncthreads();  // everybody has read the decision before the tile data overwrites it
  SweepPart sp;
  if (role < 2) {   // a unit past the whole tiles is half a sweep of one of the role's last tiles
    const int whole = (int)(role ? t_q : t_kv) - (role ? xq : xk), x = role ? xq : xk;
    if (tile >= whole) {
      const int j = tile - whole;
      sp.part = j >= x ? 1 : 0;
      const int js = j - sp.part * x;                                  // which of the split tiles
      tile = whole + js;
      const int swept = role ? (S + 127) / 128 : (Sq + 127) / 128;     // tiles of the sweep: a key-owner tile sweeps the query tiles, a query-owner tile the key tiles
      sp.t0 = sp.part ? swept / 2 : 0;
      sp.t1 = sp.part ? swept : swept / 2;
      // scratch: key-owner partials (8 DH / 32 pieces x 256 lanes x 16 B each) first, then the query-owner partials (DH / 8 pieces)
      constexpr int KV_F4 = 8 * (DH / 32) * 256, Q_F4 = (DH / 8) * 256;
      sp.scratch = split_ws + (role ? (size_t)xk * KV_F4 + (size_t)js * Q_F4 : (size_t)js * KV_F4);
      sp.flag = &ps.split_flag[role][js];
      sp.want = epoch + 1u;   // 1 .. 0x1000: never the idle 0, never another epoch's word
      sp.status = status;
    }
  }
  if (role == 0) {
    const int nt = (S + 127) / 128;
    attn_dkv2_block<128, DH, ROPE>(Q, K, V, ld, dO, ldo, LSE, delta, dK, dV, ldd, S, Sq, qoff, H, scale, tile % nt, (tile / nt) % H, tile / (nt * H), smem, sp, rs);
  } else if (role == 1) {
    const int nt = (Sq + 127) / 128;
    if constexpr (QKIND == 2 && DH == 64) attn_dq_fp8_block(f8, K, ld, LSE, delta, dQ, ldd, S, Sq, qoff, H, scale, tile % nt, (tile / nt) % H, tile / (nt * H), smem, sp);
    else if constexpr (QKIND == 1) attn_dq3_block<128, DH, ROPE>(Q, K, V, ld, dO, ldo, LSE, delta, dQ, ldd, S, Sq, qoff, H, scale, tile % nt, (tile / nt) % H, tile / (nt * H), smem, sp, rs);
    else attn_dq2_block<128, DH>(Q, K, V, ld, dO, ldo, LSE, delta, dQ, ldd, S, Sq, qoff, H, scale, tile % nt, (tile / nt) % H, tile / (nt * H), smem);
  }
  // the block that ends last (every other block has claimed its tile long ago) hands the slot over stamped with the NEXT epoch: no memset node in front of
  // the next launch (a fill kernel of ~8 us on the critical path, 16 per MIM step)
  __syncthreads();  // (the block's waves are done with the LDS the flag below travels through)
  if (threadIdx.x == 0) {
    const unsigned d = atomicAdd(&ps.done, 1u);
    if ((d >> 20) != epoch) { pair_fault(status, 0x5003u); bc[0] = 0; }
    else bc[0] = (d & PAIR_COUNT_MASK) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (bc[0]) {
    const unsigned next = ((epoch + 1u) & 0xfffu) << 20;
    if (threadIdx.x == 0) { ps.head[0] = next; ps.head[1] = next; ps.done = next; }
    for (int i = threadIdx.x; i < 2048; i += 256) ps.cu_ticket[i] = 0;   // every launch starts with "first arrival on a CU = key-owner block"
  }
}

// debug hooks (smbv_debug_pair_state, include/smbv.h): op 1 = dirty every slot's key-owner queue (what an overlapping second launch would do),
// op 2 = raise every hand-off flag of every slot the way producers of an unfinished launch of epoch `arg` would have left them,
// op 0 = every slot back to its module-load state (hand-off flags included)
__global__ void pair_state_debug_kernel(int op, unsigned arg) {
  const int slot = blockIdx.x;
  PairState& ps = g_pair_state[slot];
  if (op == 1) { if (threadIdx.x == 0) atomicAdd(&ps.head[0], 3u); return; }
  if (op == 2) { for (int i = threadIdx.x; i < 2 * PAIR_SPLIT_MAX; i += 256) ps.split_flag[0][i] = (arg & 0xfffu) + 1u; return; }
  if (threadIdx.x == 0) { ps.head[0] = 0; ps.head[1] = 0; ps.done = 0; }
  for (int i = threadIdx.x; i < 2048; i += 256) ps.cu_ticket[i] = 0;
  for (int i = threadIdx.x; i < 2 * PAIR_SPLIT_MAX; i += 256) ps.split_flag[0][i] = 0;
}

// Host side of the slots: per device, a slot belongs to the first stream that used it (at most 8 streams per device run the pair launch; further
// streams take the two-launch path, which has no shared state), and the slot's launch epoch counts that stream's launches.  A slot is therefore only
// ever touched by launches that are ordered by ONE stream: the previous launch's last block has re-stamped it before the next one starts.
struct PairHost {
  hipStream_t streams[8];
  unsigned epoch[8];
  int n = 0;
};
static PairHost g_pair_host[64];
static std::mutex g_pair_mu;

// returns the slot of (current device, st), claiming a free one for a stream seen for the first time, or -1 when all 8 slots of the device belong to other
// streams.  Does NOT consume a launch epoch: that happens in pair_epoch_take, immediately in front of the launch, after everything that can still fail.
static int pair_slot_for(hipStream_t st, int* dev_out, bool reset = false) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  std::lock_guard<std::mutex> lk(g_pair_mu);
  PairHost& h = g_pair_host[dev];
  if (reset) { h.n = 0; return 0; }
  int slot = -1;
  for (int i = 0; i < h.n; i++) if (h.streams[i] == st) { slot = i; break; }
  if (slot < 0) {
    if (h.n >= 8) return -1;
    slot = h.n++;
    h.streams[slot] = st;
    h.epoch[slot] = 0;
  }
  if (dev_out) *dev_out = dev;
  return slot;
}
// The epoch the next launch of (dev, slot) must carry; the host counter moves on.  The device slot is re-stamped with epoch + 1 by the launch's last block,
// so the counter may only advance for a launch that really is enqueued: pair_epoch_untake puts it back when the launch itself failed (an
// argument error after the epoch was consumed left host and device one apart - every later launch on that stream then faulted with 0x5001).
static unsigned pair_epoch_take(int dev, int slot) {
  std::lock_guard<std::mutex> lk(g_pair_mu);
  PairHost& h = g_pair_host[dev];
  const unsigned e = h.epoch[slot];
  h.epoch[slot] = (e + 1u) & 0xfffu;
  return e;
}
static void pair_epoch_untake(int dev, int slot, unsigned e) {
  std::lock_guard<std::mutex> lk(g_pair_mu);
  PairHost& h = g_pair_host[dev];
  if (h.epoch[slot] == ((e + 1u) & 0xfffu)) h.epoch[slot] = e;
}

int tn_split_state_debug(int op, unsigned arg, hipStream_t st);   // gemm.hip: the split grouped weight-gradient launch keeps the same kind of state

int pair_state_debug(int op, hipStream_t st) {
  // op 2 + 4 e (e = 0 .. 0xfff): stale hand-off flags of epoch e
  const int base = op & 3;
  if (op < 0 || base == 3 || (base != 2 && op != base)) { set_error("smbv_debug_pair_state: op %d (0 = reset, 1 = dirty the tile queues, 2 + 4 e = stale hand-off flags of epoch e)", op); return SMBV_ERR_ARG; }
  hipLaunchKernelGGL(pair_state_debug_kernel, dim3(8), dim3(256), 0, st, base, (unsigned)(op >> 2));
  if (base == 0) pair_slot_for(nullptr, nullptr, true);   // the host's epochs restart with the slots
  const int rc = check_launch("pair_state_debug");
  if (rc) return rc;
  return tn_split_state_debug(base, (unsigned)(op >> 2), st);
}

// how many tiles of each role the pair launch runs as two half sweeps: 192 of each (in-box A/B at the step's shapes, profiles/r05_pair_split_ab.txt: 0 / 128 / 192 /
// 256 -> 25.66 / 25.53 / 25.32 / 25.49 ms per step); tuning override: environment SMBV_PAIR_SPLIT = "x" or "xk,xq", read once; 0 = none
static void pair_split_request(int& wk, int& wq) {
  static const std::pair<int, int> v = [] {
    const char* e = getenv("SMBV_PAIR_SPLIT");
    int a = 192, b = 192;
    if (e) { a = b = atoi(e); const char* c = strchr(e, ','); if (c) b = atoi(c + 1); }
    return std::make_pair(std::max(0, std::min(a, PAIR_SPLIT_MAX)), std::max(0, std::min(b, PAIR_SPLIT_MAX)));
  }();
  wk = v.first; wq = v.second;
}
// the split of one launch: a role is split only if its sweep has at least four tiles (two per half) and it has at least twice as many tiles as are split - and
// nothing is split when the launch's tiles fill a whole number of rounds of the chip's block slots anyway (two blocks per CU at head_dim 64, three at 32): there the
// halves only add their prologues and the hand-off (ViT-large's decoder, 2,560 tiles = 5.0 rounds: +0.6 ms per step with the split, profiles/r05_pair_split_ab.txt)
void attention_bwd_split_counts(int B, int S, int Sq, int H, int Dh, bool q_splittable, int& xk, int& xq) {
  int wk, wq;
  pair_split_request(wk, wq);
  const int t_kv = cdiv(S, 128) * H * B, t_q = cdiv(Sq, 128) * H * B;
  const int slots = 256 * (Dh == 64 ? 2 : 3), last = (t_kv + t_q) % slots;
  if (last < slots / 32 && t_kv + t_q >= slots) { xk = xq = 0; return; }
  xk = (cdiv(Sq, 128) >= 4 && t_kv >= 2 * wk) ? wk : 0;
  xq = (q_splittable && cdiv(S, 128) >= 4 && t_q >= 2 * wq) ? wq : 0;
}
size_t attention_bwd_split_workspace_bytes(int Dh, int xk, int xq) {
  return ((size_t)xk * 8 * (Dh / 32) * 256 + (size_t)xq * (Dh / 8) * 256) * sizeof(f32x4);
}
// what a block of these dimensions can ask for (its launches may sweep a query sub-range: fewer tiles, never more)
size_t attention_bwd_split_workspace_max(int B, int S, int H, int Dh) {
  int xk, xq;
  int wk, wq;
  pair_split_request(wk, wq);   // (an upper bound over the launches of a block of these dimensions: a ranged launch has fewer tiles and another round count)
  const int t = cdiv(S, 128) * H * B;
  xk = t >= 2 * wk ? wk : 0;
  xq = t >= 2 * wq ? wq : 0;
  return attention_bwd_split_workspace_bytes(Dh, xk, xq);
}

static int attention_bwd_impl(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, const bf16_t* O, const bf16_t* dO, int ldo, const float* LSE,
                              float* delta, bf16_t* dQ, bf16_t* dK, bf16_t* dV, int ldd, int B, int S, int H, int Dh, float scale, int parts, int Sq, int qoff, hipStream_t st,
                              const bf16_t* O_lo, void* f8ws, size_t f8ws_bytes, void* split_ws, size_t split_ws_bytes, const RopeStore* rope, bool* rope_done);

// rope: dQ / dK are gradients of the ROTATED q / k (V-JEPA, modeling_vjepa.py:311-326) and are to leave through the transposed rotary map - in the stores of the
// one mixed-role launch (bf16 blocks on the 16x16x32 MFMA: what the steps run), bitwise what a rope3d launch over the stored matrix computes; on every other
// path (fp8 blocks, two launches, debug forms) that launch runs behind the kernels: dQ, dK must then be the q | k sections of one [B * S, ldd] matrix
int attention_bwd(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, const bf16_t* O, const bf16_t* dO, int ldo, const float* LSE,
                  float* delta, bf16_t* dQ, bf16_t* dK, bf16_t* dV, int ldd, int B, int S, int H, int Dh, float scale, int parts, int Sq, int qoff, hipStream_t st,
                  const bf16_t* O_lo, void* f8ws, size_t f8ws_bytes, void* split_ws, size_t split_ws_bytes, const RopeBwd* rope) {
  RopeStore rs;
  if (rope) {
    rs.table = rope->table; rs.pos = rope->pos;
    if (!rope->table || rope->grid <= 0 || rope->Dh <= 0 || rope->Dh > Dh || (parts & 6) != 6) { set_error("attention_bwd: rotary map needs a table, a grid, head_dim <= the slot and both gradient products"); return SMBV_ERR_ARG; }
    if (2 * ((rope->Dh / 3) / 2) <= 0) rope = nullptr;   // head_dim < 6: nothing is rotated
  }
  // in the stores only with a table whose extent is known (it is staged into LDS) and positions the division-free decode is exact for
  const bool storable = rope && rope_geo(rs.geo, rope->grid, rope->Dh, rope->rows);
  bool done = false;
  const int rc = attention_bwd_impl(Q, K, V, ld, O, dO, ldo, LSE, delta, dQ, dK, dV, ldd, B, S, H, Dh, scale, parts, Sq, qoff, st, O_lo, f8ws, f8ws_bytes, split_ws,
                                    split_ws_bytes, storable ? &rs : nullptr, &done);
  if (rc != SMBV_OK || !rope || done) return rc;
  return rope3d(dQ, ldd, (int)(dK - dQ), (long)B * S, S, H, Dh, rope->Dh, rope->grid, rope->pos, rope->table, 1, st);
}

static int attention_bwd_impl(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, const bf16_t* O, const bf16_t* dO, int ldo, const float* LSE,
                              float* delta, bf16_t* dQ, bf16_t* dK, bf16_t* dV, int ldd, int B, int S, int H, int Dh, float scale, int parts, int Sq, int qoff, hipStream_t st,
                              const bf16_t* O_lo, void* f8ws, size_t f8ws_bytes, void* split_ws, size_t split_ws_bytes, const RopeStore* rope, bool* rope_done) {
  if (Sq <= 0) { Sq = S; qoff = 0; }
  if (qoff < 0 || qoff + Sq > S) { set_error("attention_bwd: query range outside the sequence"); return SMBV_ERR_SHAPE; }
  if (Dh != 64 && Dh != 32) { set_error("attention_bwd: head_dim %d not supported (64 or 32; other widths run zero-padded in 64-wide slots, block.h)", Dh); return SMBV_ERR_SHAPE; }
  if (B <= 0 || S <= 0 || H <= 0 || (ld & 7) || (ldo & 7) || (ldd & 3)) { set_error("attention_bwd: bad shape"); return SMBV_ERR_SHAPE; }
  if ((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)O_lo | (uintptr_t)dO) & 15) || (((uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 7)) { set_error("attention_bwd: alignment"); return SMBV_ERR_ALIGN; }
  if ((size_t)(S + 256) * ld * 2 >= (1ull << 31) || (size_t)(S + 256) * ldo * 2 >= (1ull << 31)) { set_error("attention_bwd: one sequence spans >= 2 GiB of an operand matrix; the LDS-DMA offsets are 32-bit"); return SMBV_ERR_SHAPE; }
  const long pairs = (long)B * Sq * H;
  int rc = SMBV_OK;
  if (parts & 1) {
    if (Dh == 64) hipLaunchKernelGGL(attn_delta_kernel<64>, dim3(cdiv(pairs * 8, 256)), dim3(256), 0, st, O, O_lo, dO, ldo, delta, B, S, Sq, qoff, H);
    else hipLaunchKernelGGL(attn_delta_kernel<32>, dim3(cdiv(pairs * 4, 256)), dim3(256), 0, st, O, O_lo, dO, ldo, delta, B, S, Sq, qoff, H);
    rc = check_launch("attn_delta");
    if (rc) return rc;
  }
  // both products: ONE launch with mixed block roles (attn_bwd_pair_kernel: -1.3 % encoder, -2.9 % decoder, -3.7 % at head_dim 32, -6.4 % at S = 20,480 x
  // 12 heads against the two launches side by side, bitwise the same gradients: tests/test_ops_gpu.py); debug bit 524288 keeps the two launches, and so
  // does a ninth stream of one device (every slot of the scheduling state is owned by another stream)
  int dev = 0;
  const int slot = ((parts & 6) == 6 && !(g_variant & 524288)) ? pair_slot_for(st, &dev) : -1;
  if (slot >= 0) {
    // everything that can fail comes BEFORE the launch epoch is consumed: the tile-count check, the fp8 operand pre-pass (workspace size / alignment, its launches)
    // the last tiles of each role as two half sweeps (SweepPart) when the caller handed over scratch: as many as keep the launch's last round fine-grained
    int xk = 0, xq = 0;
    const int t_kv = cdiv(S, 128) * H * B, t_q = cdiv(Sq, 128) * H * B;
    if (split_ws && !((uintptr_t)split_ws & 255) && !(g_variant & 16)) {
      // (the 32x32x16 query-owner block - debug bit 2 - sweeps whole: the shipped 16x16x32 block and the fp8 block have the half-sweep form)
      const bool fp8_blocks = f8ws != nullptr && Dh == 64 && !(g_variant & 8);
      attention_bwd_split_counts(B, S, Sq, H, Dh, fp8_blocks || !(g_variant & 2), xk, xq);
      if (attention_bwd_split_workspace_bytes(Dh, xk, xq) > split_ws_bytes) xk = xq = 0;
    }
    const int blocks = t_kv + t_q + xk + xq;
    if ((unsigned)blocks >= PAIR_COUNT_MASK) { set_error("attention_bwd: %d tiles exceed the pair launch's 20-bit tile counters", blocks); return SMBV_ERR_SHAPE; }
    unsigned* status = device_status_word_dev();   // host-mapped; null if it could not be created (faults then go unreported, as before round 4)
    const bool q16 = !(g_variant & 2);   // A/B: bit 1 = the query-owner block on the 32x32x16 MFMA (attn_dq2_block)
    Fp8BwdOps f8;
    const bool fp8q = f8ws != nullptr && Dh == 64 && !(g_variant & 8);   // (debug bit 8: bf16 query-owner blocks although fp8 operands were offered)
    if (fp8q) RC(attention_bwd_fp8_prepare(Q, K, V, ld, dO, ldo, B, S, H, scale, f8ws, f8ws_bytes, st, &f8));
    const unsigned epoch = pair_epoch_take(dev, slot);
    const bool roped = rope && !fp8q && q16 && !(g_ab & 8);   // (A/B word bit 8: the rotary map as launches of its own)
    const RopeStore rs = roped ? *rope : RopeStore();
#define SMBV_PAIR_LAUNCH(DHV, KIND, ...) hipLaunchKernelGGL((attn_bwd_pair_kernel<DHV, KIND, ##__VA_ARGS__>), dim3(blocks), dim3(256), 0, st, Q, K, V, ld, dO, ldo, LSE, delta, dQ, dK, dV, ldd, B, S, Sq, qoff, H, scale, slot, epoch, status, f8, xk, xq, (f32x4*)split_ws, rs)
    if (roped && Dh == 64) SMBV_PAIR_LAUNCH(64, 1, true);
    else if (roped) SMBV_PAIR_LAUNCH(32, 1, true);
    else if (fp8q) SMBV_PAIR_LAUNCH(64, 2);
    else if (Dh == 64 && q16) SMBV_PAIR_LAUNCH(64, 1);
    else if (Dh == 64) SMBV_PAIR_LAUNCH(64, 0);
    else if (q16) SMBV_PAIR_LAUNCH(32, 1);
    else SMBV_PAIR_LAUNCH(32, 0);
#undef SMBV_PAIR_LAUNCH
    rc = check_launch("attn_bwd_pair");
    if (rc) pair_epoch_untake(dev, slot, epoch);   // nothing was enqueued: the device slot still carries this epoch
    else if (roped) *rope_done = true;
    return rc;
  }
  // dK/dV and dQ are independent (disjoint outputs, shared read-only inputs).  Their grids rarely fill a whole number of
  // block rounds (encoder: 672 blocks on 512 slots), so when both are requested they are forked onto two streams and run
  // side by side, each filling the other's idle CUs, and joined again before this function's stream continues.
  hipStream_t st_dq = st;
  ForkJoin* fj = nullptr;
  if ((parts & 6) == 6 && !(g_variant & 8192)) {
    fj = fork_join_for_current_device();
    if (fj) {
      RC(stream_edge(fj, fj->fork, st, fj->side));
      st_dq = fj->side;
    }
  }
  struct Join {  // (also on the error returns below: the forked stream never stays un-joined)
    ForkJoin* fj; hipStream_t st;
    ~Join() { if (fj) stream_edge(fj, fj->join, fj->side, st); }
  } join{fj, st};
  const int xo = ((g_variant & (1 << 25)) ? 1 : 0) | (((g_variant >> 26) & 7) << 8);  // bits 26-28: start-up stagger units (A/B)
  const bool xgrid = xo & 1;  // A/B: XCD-aware block order (smbv_common.h xcd_block_order) - measured null for the backward pair (tools/attn_ab.py 0 33554432), off
  const dim3 gkv = xgrid ? dim3(cdiv(S, 128) * H * B) : dim3(cdiv(S, 128), H, B);
  const dim3 gq = xgrid ? dim3(cdiv(Sq, 128) * H * B) : dim3(cdiv(Sq, 128), H, B);
  if (parts & 2) {
    if (Dh == 64) hipLaunchKernelGGL((attn_dkv2_kernel<128, 64>), gkv, dim3(256), 0, st, Q, K, V, ld, dO, ldo, LSE, delta, dK, dV, ldd, S, Sq, qoff, H, scale, xo);
    else hipLaunchKernelGGL((attn_dkv2_kernel<128, 32>), gkv, dim3(256), 0, st, Q, K, V, ld, dO, ldo, LSE, delta, dK, dV, ldd, S, Sq, qoff, H, scale, xo);
    rc = check_launch("attn_dkv");
    if (rc) return rc;
  }
  if (parts & 4) {
    const bool q16 = !(g_variant & 2);
    if (Dh == 64 && q16) hipLaunchKernelGGL((attn_dq2_kernel<128, 64, true>), gq, dim3(256), 0, st_dq, Q, K, V, ld, dO, ldo, LSE, delta, dQ, ldd, S, Sq, qoff, H, scale, xo);
    else if (Dh == 64) hipLaunchKernelGGL((attn_dq2_kernel<128, 64, false>), gq, dim3(256), 0, st_dq, Q, K, V, ld, dO, ldo, LSE, delta, dQ, ldd, S, Sq, qoff, H, scale, xo);
    else if (q16) hipLaunchKernelGGL((attn_dq2_kernel<128, 32, true>), gq, dim3(256), 0, st_dq, Q, K, V, ld, dO, ldo, LSE, delta, dQ, ldd, S, Sq, qoff, H, scale, xo);
    else hipLaunchKernelGGL((attn_dq2_kernel<128, 32, false>), gq, dim3(256), 0, st_dq, Q, K, V, ld, dO, ldo, LSE, delta, dQ, ldd, S, Sq, qoff, H, scale, xo);
    rc = check_launch("attn_dq");
  }
  return rc;  // ~Join: `st` waits for the dQ kernel
}

}  // namespace smbv
