// Attention forward with OCP e4m3 operands on the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, 2x the bf16 matrix rate):
// BASELINE.json config 5 ("ViT-large MIM, fp8 MFMA QK^T / PV").  Same function as attention_fwd.hip - softmax(Q K^T * scale) V per head,
// head_dim 64, fp32 online softmax, LSE out (reference src/models/videomae/modeling_videomae.py:196-223) - with Q, K, V and P rounded to
// e4m3 under power-of-two (E8M0) block scales, MX style:
//   * quant_qk_fp8_kernel: Q (pre-multiplied by scale * log2 e, so scores leave the MFMA in the exp2 domain) and K, one scale per
//     (token, head, 32 channels) - the 32 contraction elements one lane feeds the MFMA;
//   * quant_vt_fp8_kernel: V transposed to [batch, head, channel][key] with one scale per (channel, 32 consecutive keys), the keys of every
//     64-key group stored in the order the P operand holds them (below), so both operands of P V are plain 32-byte row reads;
//   * attn_fwd_fp8_kernel: 4 waves x 32 queries, 64-key tiles.  S^T = K Q^T is ONE MFMA per 32 keys (K = 64 channels = the two lane halves
//     x 32 bytes), query on the lane: row max / sum are lane-local plus one v_permlane32_swap; the exponentiated accumulator, scaled by 2^7
//     and converted to e4m3 in register order, IS the B operand of O^T += V^T P^T (its E8M0 scale 2^-7 undoes the factor) - lane half h of the
//     two S^T tiles holds keys 32 t + 8 g + 4 h + e (t tile, g = 0..3, e = 0..3) in bytes 16 t + 4 g + e, and V^T is stored in that key order.
// The operand maps were pinned on hardware by tools/probe/fp8_probe.hip: lane (r = l & 31, h = l >> 5) feeds row / column r and 32 of the
// 64 contraction elements; byte j of half h meets byte j of the other operand's half h; the E8M0 scale in lane (r, h) acts on scale block h of
// row r, and scale block b is bytes 16 b .. 16 b + 15 of BOTH lane halves.  So lane half h carries channels 16 h .. +15 (block 0) and
// 32 + 16 h .. +15 (block 1) of a Q / K row, and for P V block t is simply the keys of S^T tile t.
// The backward stays on the bf16 kernels (attention_bwd.hip) with this forward's LSE.  Stated tolerance: tests/test_fp8_gpu.py.
#include "smbv_common.h"
#include "smbv_kernels.h"

namespace smbv {

typedef int v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int e8m0_for(float amax) {
  // smallest power of two 2^e with amax / 2^e <= 448 (e4m3 max); stored biased by 127
  if (!(amax > 0.f)) return 127;
  const int e = (int)ceilf(__log2f(amax * (1.0f / 448.0f)));
  return min(max(e, -126), 127) + 127;
}
__device__ __forceinline__ float e8m0_inv(int byte) { return __uint_as_float((unsigned)(254 - byte) << 23); }  // 2^-(byte-127)

// bf16 [rows, ld] column block `col0 .. col0 + w` -> e4m3 [rows, w] + E8M0 scales [rows, w / 32]; premul folded in before rounding
// scale layout: head_major_S == 0 -> [rows, w / 32]; > 0 (= S) -> [batch, head][S][2], contiguous per head (what the K tiles are staged from)
__global__ __launch_bounds__(256) void quant_qk_fp8_kernel(const bf16_t* __restrict__ X, int ld, long rows, int w, float premul,
                                                           unsigned char* __restrict__ X8, unsigned char* __restrict__ sX, int head_major_S) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = w >> 3;
  const long row = gid / per_row;
  const int c8 = (int)(gid % per_row);
  float v[8];
  float amax = 0.f;
  if (row < rows) {
    const u32x4 raw = *(const u32x4*)(X + row * ld + c8 * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) { v[2 * i] = bf_lo(raw[i]) * premul; v[2 * i + 1] = bf_hi(raw[i]) * premul; }
#pragma unroll
    for (int i = 0; i < 8; i++) amax = fmaxf(amax, fabsf(v[i]));
  }
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));  // 4 threads = 32 channels = one scale block
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  if (row >= rows) return;
  const int sb = e8m0_for(amax);
  const float inv = e8m0_inv(sb);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
  *(u32x2*)(X8 + row * w + c8 * 8) = (u32x2){(unsigned)lo, (unsigned)hi};
  if ((c8 & 3) == 0) {
    if (head_major_S > 0) {
      const long bb = row / head_major_S, ss = row % head_major_S;
      const int H = w >> 6, head = c8 >> 3, half = (c8 >> 2) & 1;
      sX[((bb * H + head) * head_major_S + ss) * 2 + half] = (unsigned char)sb;
    } else {
      sX[row * (w >> 5) + (c8 >> 2)] = (unsigned char)sb;
    }
  }
}

// V bf16 [B*S, ld] (head h at columns 64 h ..) -> V8T e4m3 [B, H, 64 channels][Spad keys, permuted inside 64-key groups] + scales
// sV [B, H][Spad / 64][64][2] (contiguous per 64-key group).  One block = one (64-key group, head, batch); thread (d = t >> 1, half = t & 1) owns one scale block.
__global__ __launch_bounds__(128) void quant_vt_fp8_kernel(const bf16_t* __restrict__ V, int ld, int S, int Spad, int H,
                                                           unsigned char* __restrict__ V8T, unsigned char* __restrict__ sV) {
  __shared__ unsigned short tile[64][66];
  const int t = threadIdx.x, grp = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  for (int i = t; i < 64 * 8; i += 128) {  // 64 keys x 8 chunks of 8 channels
    const int key = i >> 3, c = i & 7;
    const int gk = grp * 64 + key;
    u32x4 raw = {0, 0, 0, 0};
    if (gk < S) raw = *(const u32x4*)(V + ((size_t)b * S + gk) * ld + head * 64 + c * 8);
#pragma unroll
    for (int j = 0; j < 4; j++) { tile[key][c * 8 + 2 * j] = (unsigned short)(raw[j] & 0xffff); tile[key][c * 8 + 2 * j + 1] = (unsigned short)(raw[j] >> 16); }
  }
  __syncthreads();
  const int d = t >> 1, tt = t & 1;  // channel, 32-key scale block (= S^T tile) of the group
  float v[32];
  float amax = 0.f;
#pragma unroll
  for (int kk = 0; kk < 32; kk++) {
    v[kk] = bf2f(tile[32 * tt + kk][d]);
    amax = fmaxf(amax, fabsf(v[kk]));
  }
  const int sb = e8m0_for(amax);
  const float inv = e8m0_inv(sb);
  // lane half h of the P operand holds keys 32 tt + 8 g + 4 h + e in bytes 16 tt + 4 g + e: four dwords per (block, half)
  unsigned char* dst = V8T + (((size_t)b * H + head) * 64 + d) * Spad + (size_t)grp * 64 + 16 * tt;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    unsigned out[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      int wv = 0;
      wv = __builtin_amdgcn_cvt_pk_fp8_f32(v[8 * g + 4 * h] * inv, v[8 * g + 4 * h + 1] * inv, wv, false);
      wv = __builtin_amdgcn_cvt_pk_fp8_f32(v[8 * g + 4 * h + 2] * inv, v[8 * g + 4 * h + 3] * inv, wv, true);
      out[g] = (unsigned)wv;
    }
    *(u32x4*)(dst + 32 * h) = (u32x4){out[0], out[1], out[2], out[3]};
  }
  sV[((((size_t)b * H + head) * (Spad >> 6) + grp) * 64 + d) * 2 + tt] = (unsigned char)sb;
}

// value of the partner lane (l ^ 32) combined with this lane's: one v_permlane32_swap instead of an LDS-routed shuffle
__device__ __forceinline__ void other_half(float v, float& own, float& other) {
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  const unsigned a = __builtin_bit_cast(unsigned, v);
  const u32x2v sw = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  unsigned lo = sw[0], hi = sw[1];
  asm volatile("" : "+v"(lo), "+v"(hi));  // (hipcc -O3 otherwise folds a symmetric use of sw[0], sw[1] to sw[0]: attention_fwd.hip)
  own = __builtin_bit_cast(float, lo);
  other = __builtin_bit_cast(float, hi);
}

__device__ __forceinline__ int fp8_swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(const unsigned char* __restrict__ wsb, unsigned ws_bytes, unsigned offK, unsigned offSQ,
                                                              unsigned offSK, unsigned offV, unsigned offSV, int w, bf16_t* __restrict__ O, int ldo,
                                                              float* __restrict__ LSE, int S, int Spad, int Sq, int qoff, int H) {
  // K / V^T / scale tiles go global -> LDS by LDS-DMA (hidden from hipcc's wait insertion, smbv_common.h) into a ring of three slots, two tiles
  // in flight; one counted vmcnt + one barrier per tile.  Slot: K tile [64 keys][64 B] | V^T tile [64 channels][64 B] | per 16 rows
  // [K scales 32 B | V scales 32 B] x 4.  (A one-tile register prefetch left every tile waiting for its L2 round trip.)
  constexpr int SLOT = 4096 + 4096 + 256;
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * SLOT];
  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = l & 31, h = l >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * 128 + wv * 32 + r;            // index inside the query range
  const bool qok = q < Sq;
  const size_t qrow = (size_t)b * S + qoff + (qok ? q : 0);
  const int nsc = w >> 5;                                   // Q scale bytes per token row
  const unsigned char* Q8 = wsb;
  v8i qf = {0, 0, 0, 0, 0, 0, 0, 0};
  int qs = 127;
  if (qok) {
    const u32x4 a = *(const u32x4*)(Q8 + qrow * w + head * 64 + 16 * h);        // channels 16 h .. (scale block 0)
    const u32x4 c = *(const u32x4*)(Q8 + qrow * w + head * 64 + 32 + 16 * h);   // channels 32 + 16 h .. (scale block 1)
    qf = (v8i){(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)c[0], (int)c[1], (int)c[2], (int)c[3]};
    qs = (wsb + offSQ)[qrow * nsc + head * 2 + h];
  }
  asm volatile("" : "+v"(qf), "+v"(qs));  // consumed here: hipcc waits for these loads now, not with a vmcnt(0) inside the tile loop
  // copy side: wave wv stages rows 16 wv .. + 15 of both tiles (one 1-KiB piece each: lane -> row l >> 2, physical chunk l & 3 holding logical
  // chunk (l & 3) ^ ((row >> 2) & 3)) and, with its first 16 lanes, the 64 scale bytes of those rows
  const i32x4 rs = dma_rsrc(wsb, ws_bytes);
  const int prow = 16 * wv + (l >> 2), csrc = (l & 3) ^ ((l >> 4) & 3);
  const int ok0 = (int)offK + (int)(((size_t)b * S + prow) * w) + head * 64 + 16 * csrc;
  const int ov0 = (int)offV + (int)((((size_t)b * H + head) * 64 + prow) * Spad) + 16 * csrc;
  const int j8 = l & 7;
  const int os0 = l < 8 ? (int)offSK + (int)((((size_t)b * H + head) * S + 16 * wv) * 2) + 4 * j8
                        : (int)offSV + (int)((((size_t)b * H + head) * (Spad >> 6) * 64 + 16 * wv) * 2) + 4 * j8;
  const int os_step = l < 8 ? 128 : 128;  // both scale arrays advance 64 rows x 2 B per tile
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((int)lds_addr_of(smem));
  auto stage = [&](int kt) __attribute__((always_inline)) {
    const unsigned sl = lds0 + (unsigned)(kt % 3) * SLOT;
    dma16_hidden(rs, sl + (unsigned)wv * 1024, ok0 + kt * 64 * w);
    dma16_hidden(rs, sl + 4096 + (unsigned)wv * 1024, ov0 + kt * 64);
    if (l < 16) dma4_hidden(rs, sl + 8192 + (unsigned)wv * 64, os0 + kt * os_step);
  };

  f32x16 o0, o1;
#pragma unroll
  for (int i = 0; i < 16; i++) { o0[i] = 0.f; o1[i] = 0.f; }
  // VALU diet (the loop is bound by vector issue, not by the matrix pipe, at 64 channels per head): scores leave the MFMA already relative to a
  // running per-query reference `mref` and shifted by +7 (the reference sits in the accumulator input), so p' = 128 exp2(s - mref) is ONE
  // v_exp_f32 per score with no subtract and no multiply, and feeds the e4m3 conversion directly; the reference only moves when a score would
  // push p' past 256 (or on the first tile) - the rescale of O and of the row sum then runs under a wave-uniform branch, rarely.
  // (the loop is vector-issue bound - SQ_ACTIVE_INST_VALU x 3 waves per SIMD = 0.9 - so two more things leave the vector pipe: the reference
  // lives in a persistent accumulator-input block `cin` instead of being splatted per tile, and the row sum is ONE more MFMA, ones^T P^T, on the
  // e4m3 P the numerator uses, instead of 32 adds)
  float mref = 0.f;
  f32x16 cin, ol;
#pragma unroll
  for (int i = 0; i < 16; i++) { cin[i] = 7.0f; ol[i] = 0.f; }
  const v8i ones = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};  // e4m3 1.0
  const int nt = (S + 63) >> 6;
  int pw[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the tile's 64 probabilities as e4m3 (loop-carried only as the conversions' pass-through operand)
  stage(0);
  stage(1);
  for (int kt = 0; kt < nt; kt++) {
    // this wave's pieces of tile kt have landed (tile kt + 1 may still fly: three copies a tile, the wave with lanes >= 16 idle in the third
    // still counts it); the barrier publishes the tile and retires everybody's reads of tile kt - 1, whose slot tile kt + 2 now takes
    // (past the end the copies read unused bytes of the workspace: never consumed)
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage(kt + 2);
    const unsigned char* sb = smem + (kt % 3) * SLOT;
    const unsigned char* scb = sb + 8192;
    f32x16 s[2];
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      const int row = 32 * tt + r;
      const u32x4 a = *(const u32x4*)(sb + fp8_swz(row, h));
      const u32x4 c = *(const u32x4*)(sb + fp8_swz(row, 2 + h));
      const v8i kf = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)c[0], (int)c[1], (int)c[2], (int)c[3]};
      const int ks = scb[(row >> 4) * 64 + (row & 15) * 2 + h];
      s[tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf, cin, 0, 0, 0, ks, 0, qs);   // S^T[key][q] - mref + 7, exp2 domain
    }
    if (kt == nt - 1 && (S & 63)) {  // keys past the end of the sequence
#pragma unroll
      for (int tt = 0; tt < 2; tt++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int key = kt * 64 + 32 * tt + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= S) s[tt][i] = -INFINITY;
        }
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(fmaxf(mx, s[0][i]), s[1][i]);  // v_max3_f32
    {  // the other lane half holds the other 32 keys of this query
      float a, c2;
      other_half(mx, a, c2);
      mx = fmaxf(a, c2);
    }
    const bool rebase = kt == 0 || mx > 8.0f;
    if (__builtin_amdgcn_ballot_w64(rebase) != 0) {
      const float d = rebase ? mx - 7.0f : 0.f;  // bring this query's maximum back to 2^7
      const float alpha = fast_exp2(-d);
      mref += d;
#pragma unroll
      for (int i = 0; i < 16; i++) { o0[i] *= alpha; o1[i] *= alpha; ol[i] *= alpha; s[0][i] -= d; s[1][i] -= d; cin[i] -= d; }
    }
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const float p0 = fast_exp2(s[tt][4 * g]), p1 = fast_exp2(s[tt][4 * g + 1]), p2 = fast_exp2(s[tt][4 * g + 2]), p3 = fast_exp2(s[tt][4 * g + 3]);
        int wd = pw[4 * tt + g];   // (both halves are overwritten: the previous tile's word as the pass-through operand saves a v_mov 0 per word)
        wd = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, wd, false);   // p' in [0, 256]: 7-8 binades above where P in [0, 1] would sit in e4m3
        wd = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, wd, true);
        pw[4 * tt + g] = wd;
      }
    const v8i pf = {pw[0], pw[1], pw[2], pw[3], pw[4], pw[5], pw[6], pw[7]};
    ol = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, pf, ol, 0, 0, 0, 127, 0, 127);   // every row: sum over the tile's 64 keys of p' (as rounded)
#pragma unroll
    for (int dt = 0; dt < 2; dt++) {
      const int row = 32 * dt + r;  // channel
      const u32x4 a = *(const u32x4*)(sb + 4096 + fp8_swz(row, 2 * h));
      const u32x4 c = *(const u32x4*)(sb + 4096 + fp8_swz(row, 2 * h + 1));
      const v8i vf = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)c[0], (int)c[1], (int)c[2], (int)c[3]};
      const int vs = scb[(row >> 4) * 64 + 32 + (row & 15) * 2 + h];
      if (dt == 0) o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, o0, 0, 0, 0, vs, 0, 120);   // O^T[channel][q]; 120 = 2^-7 undoes the 128
      else o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, o1, 0, 0, 0, vs, 0, 120);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead copies still target this block's LDS
  const float ltot = ol[0];   // = 128 * sum exp2(s - mref) over ALL keys (the MFMA contracts both lane halves)
  if (!qok) return;
  const float inv = 128.0f / ltot;
  bf16_t* op = O + qrow * ldo + head * 64;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int d0 = 8 * g + 4 * h;
    const u32x2 a = {pack_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv)};
    const u32x2 c = {pack_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv)};
    *(u32x2*)(op + d0) = a;
    *(u32x2*)(op + 32 + d0) = c;
  }
  if (h == 0) LSE[((size_t)b * H + head) * Sq + q] = (mref + __log2f(ltot) - 7.0f) * 0.6931471805599453f;
}

// four row-quantisation passes of the same shape as ONE launch (blockIdx.y = pass): the backward's Q, K, V, dO
struct QuantJob { const bf16_t* X; int ld; float premul; unsigned char* X8; unsigned char* sX; int head_major_S; };
struct QuantJobs4 { QuantJob j[4]; };
__global__ __launch_bounds__(256) void quant_rows4_fp8_kernel(QuantJobs4 jobs, long rows, int w) {
  const QuantJob& J = jobs.j[blockIdx.y];
  const bf16_t* __restrict__ X = J.X;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = w >> 3;
  const long row = gid / per_row;
  const int c8 = (int)(gid % per_row);
  float v[8];
  float amax = 0.f;
  if (row < rows) {
    const u32x4 raw = *(const u32x4*)(X + row * J.ld + c8 * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) { v[2 * i] = bf_lo(raw[i]) * J.premul; v[2 * i + 1] = bf_hi(raw[i]) * J.premul; }
#pragma unroll
    for (int i = 0; i < 8; i++) amax = fmaxf(amax, fabsf(v[i]));
  }
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));  // 4 threads = 32 channels = one scale block
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  if (row >= rows) return;
  const int sb = e8m0_for(amax);
  const float inv = e8m0_inv(sb);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
  *(u32x2*)(J.X8 + row * w + c8 * 8) = (u32x2){(unsigned)lo, (unsigned)hi};
  if ((c8 & 3) == 0) {
    if (J.head_major_S > 0) {
      const long bb = row / J.head_major_S, ss = row % J.head_major_S;
      const int H = w >> 6, head = c8 >> 3, half = (c8 >> 2) & 1;
      J.sX[((bb * H + head) * J.head_major_S + ss) * 2 + half] = (unsigned char)sb;
    } else {
      J.sX[row * (w >> 5) + (c8 >> 2)] = (unsigned char)sb;
    }
  }
}

// ---- operands of the fp8 BACKWARD (attention_bwd.hip attn_dq_fp8_block: the query-owner blocks' S'^T = K Q^T and dP'^T = V dO^T recomputation on the
// block-scaled fp8 MFMA): Q (pre-scaled into the exp2 domain), K, V and dO as e4m3 ROWS with one E8M0 scale per (token, head, 32 channels) - the same
// quantisation pass the forward uses for Q and K (V here by rows: the contraction of dP is over channels, not keys).  K and V scales head-major
// ([batch, head][S][2]: what the tile copies stage), Q and dO scales row-major.
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t attention_bwd_fp8_workspace_bytes(int B, int S, int H) {
  const size_t rows = (size_t)B * S, w = (size_t)H * 64;
  return 4 * al256(rows * w) + 4 * al256(rows * (w >> 5)) + 256;   // (+ 256: the run-ahead tile copies past the last tile read inside the buffer)
}
int attention_bwd_fp8_prepare(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, const bf16_t* dO, int ldo, int B, int S, int H, float scale, void* ws,
                              size_t ws_bytes, hipStream_t st, Fp8BwdOps* ops) {
  const size_t need = attention_bwd_fp8_workspace_bytes(B, S, H);
  if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) { set_error("attention_bwd (fp8): workspace too small or misaligned (%zu < %zu)", ws_bytes, need); return SMBV_ERR_WORKSPACE; }
  if (need >= ((size_t)1 << 31)) { set_error("attention_bwd (fp8): workspace of %zu bytes exceeds the 32-bit copy offsets", need); return SMBV_ERR_SHAPE; }
  const long rows = (long)B * S;
  const int w = H * 64;
  unsigned char* p = (unsigned char*)ws;
  unsigned char* Q8 = p; p += al256((size_t)rows * w);
  unsigned char* K8 = p; p += al256((size_t)rows * w);
  unsigned char* V8 = p; p += al256((size_t)rows * w);
  unsigned char* D8 = p; p += al256((size_t)rows * w);
  unsigned char* sQ = p; p += al256((size_t)rows * (w >> 5));
  unsigned char* sK = p; p += al256((size_t)rows * (w >> 5));
  unsigned char* sV = p; p += al256((size_t)rows * (w >> 5));
  unsigned char* sD = p;
  const int qblocks = cdiv(rows * (w >> 3), 256);
  QuantJobs4 jobs;
  jobs.j[0] = {Q, ld, scale * 1.4426950408889634f, Q8, sQ, 0};
  jobs.j[1] = {K, ld, 1.0f, K8, sK, S};
  jobs.j[2] = {V, ld, 1.0f, V8, sV, S};
  jobs.j[3] = {dO, ldo, 1.0f, D8, sD, 0};
  hipLaunchKernelGGL(quant_rows4_fp8_kernel, dim3(qblocks, 4), dim3(256), 0, st, jobs, rows, w);   // one launch: four passes of ~15 MB each
  int rc = check_launch("quant_fp8 (backward)");
  if (rc) return rc;
  const unsigned char* base = (const unsigned char*)ws;
  ops->base = base; ops->bytes = (unsigned)need; ops->w = w;
  ops->offK = (unsigned)(K8 - base); ops->offV = (unsigned)(V8 - base); ops->offD = (unsigned)(D8 - base);
  ops->offSQ = (unsigned)(sQ - base); ops->offSK = (unsigned)(sK - base); ops->offSV = (unsigned)(sV - base); ops->offSD = (unsigned)(sD - base);
  return SMBV_OK;
}

size_t attention_fp8_workspace_bytes(int B, int S, int H) {
  const size_t rows = (size_t)B * S, w = (size_t)H * 64, Spad = ((size_t)S + 63) & ~(size_t)63;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return al(rows * w) * 2 + al(rows * (w >> 5)) * 2 + al((size_t)B * H * 64 * Spad) + al((size_t)B * H * 64 * (Spad >> 6) * 2);
}

int attention_fwd_fp8(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int ld, bf16_t* O, int ldo, float* LSE, int B, int S, int H, int Dh, float scale,
                      int Sq, int qoff, void* ws, size_t ws_bytes, hipStream_t st) {
  if (Sq <= 0) { Sq = S; qoff = 0; }
  if (qoff < 0 || qoff + Sq > S) { set_error("attention_fwd_fp8: query range outside the sequence"); return SMBV_ERR_SHAPE; }
  if (Dh != 64) { set_error("attention_fwd_fp8: head_dim %d not supported (only 64)", Dh); return SMBV_ERR_SHAPE; }
  if (B <= 0 || S <= 0 || H <= 0 || (ld & 7) || (ldo & 3)) { set_error("attention_fwd_fp8: bad shape"); return SMBV_ERR_SHAPE; }
  if ((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) || ((uintptr_t)O & 7)) { set_error("attention_fwd_fp8: alignment"); return SMBV_ERR_ALIGN; }
  const size_t need = attention_fp8_workspace_bytes(B, S, H);
  if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) { set_error("attention_fwd_fp8: workspace too small or misaligned (%zu < %zu)", ws_bytes, need); return SMBV_ERR_WORKSPACE; }
  const long rows = (long)B * S;
  const int w = H * 64, Spad = (S + 63) & ~63;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  unsigned char* p = (unsigned char*)ws;
  unsigned char* Q8 = p; p += al((size_t)rows * w);
  unsigned char* K8 = p; p += al((size_t)rows * w);
  unsigned char* sQ = p; p += al((size_t)rows * (w >> 5));
  unsigned char* sK = p; p += al((size_t)rows * (w >> 5));
  unsigned char* V8T = p; p += al((size_t)B * H * 64 * Spad);
  unsigned char* sV = p;
  const int qblocks = cdiv(rows * (w >> 3), 256);
  if (need >= ((size_t)1 << 31)) { set_error("attention_fwd_fp8: workspace of %zu bytes exceeds the 32-bit copy offsets", need); return SMBV_ERR_SHAPE; }
  hipLaunchKernelGGL(quant_qk_fp8_kernel, dim3(qblocks), dim3(256), 0, st, Q, ld, rows, w, scale * 1.4426950408889634f, Q8, sQ, 0);
  hipLaunchKernelGGL(quant_qk_fp8_kernel, dim3(qblocks), dim3(256), 0, st, K, ld, rows, w, 1.0f, K8, sK, S);
  hipLaunchKernelGGL(quant_vt_fp8_kernel, dim3(Spad / 64, H, B), dim3(128), 0, st, V, ld, S, Spad, H, V8T, sV);
  int rc = check_launch("quant_fp8");
  if (rc) return rc;
  const unsigned char* base = (const unsigned char*)ws;
  hipLaunchKernelGGL(attn_fwd_fp8_kernel, dim3(cdiv(Sq, 128), H, B), dim3(256), 0, st, base, (unsigned)need, (unsigned)(K8 - base), (unsigned)(sQ - base),
                     (unsigned)(sK - base), (unsigned)(V8T - base), (unsigned)(sV - base), w, O, ldo, LSE, S, Spad, Sq, qoff, H);
  return check_launch("attention_fwd_fp8");
}

}  // namespace smbv
