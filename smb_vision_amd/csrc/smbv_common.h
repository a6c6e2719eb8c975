// Shared device helpers for the smbv HIP kernels (gfx950 / CDNA4 only: 64-wide waves, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef uint16_t bf16_t;  // storage type for bf16 in global memory

#define SMBV_OK 0
#define SMBV_ERR_SHAPE (-1)
#define SMBV_ERR_ALIGN (-2)
#define SMBV_ERR_ARG (-3)
#define SMBV_ERR_WORKSPACE (-4)
#define SMBV_ERR_STATE (-5)

namespace smbv {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// One pair of the reference's 3-axis rotary map (modeling_vjepa.py:204-228; vjepa_ops.hip rope3d_kernel) or of its transpose, with the contraction spelled
// out: the map is applied by the stand-alone kernel, by the QKV product's epilogue and by the attention backward's stores, and all three must round alike
__device__ __forceinline__ void rope_pair(float x0, float x1, float c0, float s0, float c1, float s1, bool transpose, float& y0, float& y1) {
  if (!transpose) { y0 = __builtin_fmaf(x0, c0, -(x1 * s0)); y1 = __builtin_fmaf(x1, c1, x0 * s1); }   // out = x*cos + rot(x)*sin, rot = (-x1, x0)
  else { y0 = __builtin_fmaf(x0, c0, x1 * s1); y1 = __builtin_fmaf(x1, c1, -(x0 * s0)); }              // transposed map for the gradient
}
// what a lane needs of the map for ITS columns (4 consecutive elements e0 .. e0 + 3 of a head, e0 a multiple of 4: two pairs) - token-independent.
// No integer division: e < 3 seg, and an element t of a segment uses angle t % (seg / 2) with t < seg (the reference TILES cos / sin)
struct RopeLane {
  int on[2];        // pair j is rotated (inside the rotated width 3 * seg)
  int axis[2];      // 0 frame, 1 height, 2 width
  int i0[2], i1[2]; // table columns of the pair's two angles
};
__device__ __forceinline__ RopeLane rope_lane(int e0, int seg, bool in_section) {
  RopeLane r;
  const int half = seg >> 1;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int e = e0 + 2 * j;
    r.on[j] = in_section && e < 3 * seg;
    const int ax = (e >= seg ? 1 : 0) + (e >= 2 * seg ? 1 : 0);
    const int t0 = r.on[j] ? e - ax * seg : 0, t1 = t0 + 1;   // (seg is even and e is even: t1 < seg; a pair outside the width gets valid indices it will not use)
    r.axis[j] = ax;
    r.i0[j] = t0 >= half ? t0 - half : t0;
    r.i1[j] = t1 >= half ? t1 - half : t1;
  }
  return r;
}
// a token's three position ids -> its three table rows (element offsets into the table)
struct RopeTok { int row[3]; };
__device__ __forceinline__ RopeTok rope_tok(int p, int grid, int seg) {
  const int tpf = grid * grid;
  const int frame = p / tpf, rem = p - tpf * frame, height = rem / grid, width = rem - grid * height;
  return RopeTok{{frame * seg, height * seg, width * seg}};
}
// the same without integer division, for the fused forms (16 tokens per lane in a GEMM epilogue: two divisions per token were a third of that epilogue):
// m = ceil(2^32 / d), q = umulhi(p, m) is exact while p * d < 2^32 (rope_geo checks it on the host against the table's row count)
struct RopeGeo {
  int grid = 0, seg = 0, rows = 0;        // rows: positions the table covers per axis (the table is staged into LDS: rows * seg floats)
  unsigned m_tpf = 0, m_grid = 0;
  int S = 0; unsigned m_S = 0;            // row -> position inside its sequence, for callers without position ids (rope_geo_rows)
};
__device__ __forceinline__ int rope_row_pos(int m, const RopeGeo& g) { return g.S == 1 ? 0 : m - (int)__umulhi((unsigned)m, g.m_S) * g.S; }
__device__ __forceinline__ RopeTok rope_tok(int p, const RopeGeo& g) {
  const unsigned up = (unsigned)p, tpf = (unsigned)(g.grid * g.grid);
  const unsigned frame = g.grid == 1 ? up : __umulhi(up, g.m_tpf), rem = up - frame * tpf;
  const unsigned height = g.grid == 1 ? rem : __umulhi(rem, g.m_grid), width = rem - height * (unsigned)g.grid;
  return RopeTok{{(int)frame * g.seg, (int)height * g.seg, (int)width * g.seg}};
}
static inline bool rope_geo(RopeGeo& g, int grid, int Dh, int table_rows) {   // false: the fused forms cannot take this map (the stand-alone launch can)
  g.grid = grid; g.seg = 2 * ((Dh / 3) / 2); g.rows = table_rows;
  const unsigned long long tpf = (unsigned long long)grid * grid;
  if (grid <= 0 || g.seg <= 0 || table_rows <= 0 || (size_t)table_rows * g.seg * sizeof(float) > 32768 || tpf * tpf * (unsigned long long)table_rows >= (1ull << 32)) return false;
  g.m_tpf = tpf > 1 ? (unsigned)(((1ull << 32) + tpf - 1) / tpf) : 0;
  g.m_grid = grid > 1 ? (unsigned)(((1ull << 32) + grid - 1) / grid) : 0;
  return true;
}
static inline bool rope_geo_rows(RopeGeo& g, long rows, int S) {   // row % S by multiplication: exact while rows * S < 2^32
  if (S <= 0 || (unsigned long long)rows * (unsigned long long)S >= (1ull << 32)) return false;
  g.S = S; g.m_S = S > 1 ? (unsigned)(((1ull << 32) + S - 1) / S) : 0;
  return true;
}
// block-cooperative copy of the table into LDS (the caller brackets it with barriers)
__device__ __forceinline__ void rope_stage_table(float* lds, const float* __restrict__ table, const RopeGeo& g, int tid, int nthreads) {
  const int nf = g.rows * g.seg;
  for (int i = tid; i < nf; i += nthreads) lds[i] = table[i];
}
// the map (or its transpose) on 4 consecutive elements of one token, already rounded to bf16 where the stand-alone kernel would have read bf16
__device__ __forceinline__ void rope_apply4(f32x4& v, const RopeLane& rl, const RopeTok& tk, int seg, const float* __restrict__ table, bool transpose) {
  // branch-free (a pair outside the rotated width reads valid angles - rope_lane clamps - and keeps its values): the reads of all pairs and of the caller's
  // other tokens issue back to back instead of one exec-masked block with its own wait per pair
  const int half = seg >> 1;
  float c0[2], s0[2], c1[2], s1[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const float* tb = table + (rl.axis[j] == 0 ? tk.row[0] : (rl.axis[j] == 1 ? tk.row[1] : tk.row[2]));
    c0[j] = tb[rl.i0[j]]; s0[j] = tb[half + rl.i0[j]]; c1[j] = tb[rl.i1[j]]; s1[j] = tb[half + rl.i1[j]];
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    float y0, y1;
    rope_pair(v[2 * j], v[2 * j + 1], c0[j], s0[j], c1[j], s1[j], transpose, y0, y1);
    v[2 * j] = rl.on[j] ? y0 : v[2 * j]; v[2 * j + 1] = rl.on[j] ? y1 : v[2 * j + 1];
  }
}
// round to bf16, apply, ready for the final pack (what "rope3d on the stored bf16 matrix" computes)
__device__ __forceinline__ void rope_round_apply4(f32x4& v, const RopeLane& rl, const RopeTok& tk, int seg, const float* __restrict__ table, bool transpose) {
  const unsigned b0 = pack_bf16(v[0], v[1]), b1 = pack_bf16(v[2], v[3]);
  v = (f32x4){bf_lo(b0), bf_hi(b0), bf_lo(b1), bf_hi(b1)};
  rope_apply4(v, rl, tk, seg, table, transpose);
}
// the map as the attention backward's stores take it (attention_bwd.hip, ROPE builds): dQ / dK leave through the TRANSPOSED map
struct RopeStore {
  const float* table = nullptr;
  const int* pos = nullptr;   // token position ids by row of the [B * S] token matrix (null: the token's index inside its sequence)
  RopeGeo geo;
};

// transposed LDS read (verified by tools/probe/hw_probe.hip): per 16-lane group a 4x16 block of 16-bit
// elements; lane 4q+p supplies &block[q][4p]; lane i receives {block[0][i], block[1][i], block[2][i], block[3][i]}.
__device__ __forceinline__ s16x4 lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lds_addr);
}

// ---- LDS-DMA hidden from hipcc's wait-count bookkeeping -------------------------------------------------------------------------
// hipcc waits vmcnt(0) before any LDS read that follows an LDS-DMA it knows about (it cannot prove the two do not alias), which drains a
// multi-tile ring at every tile.  Issued from an asm statement the copy is invisible to that pass; the kernel then owns the protocol:
// a counted `s_waitcnt vmcnt(N)` by every issuing wave, then a barrier, then the reads.
typedef int i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor (stride 0, `bytes` addressable; accesses past it read zeros), forced into SGPRs
__device__ __forceinline__ i32x4 dma_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffull));
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffull));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
// one wave-instruction: lane i copies 16 B from (descriptor base + voff) to LDS byte address lds_uniform + 16*i
__device__ __forceinline__ void dma16_hidden(i32x4 rsrc, unsigned lds_uniform, int voff) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_uniform), "v"(voff), "s"(rsrc)
               : "memory");
}

// the same with 4 bytes per lane (256 B per wave-instruction; honours the EXEC mask: lane i of the active lanes writes lds_uniform + 4*i)
__device__ __forceinline__ void dma4_hidden(i32x4 rsrc, unsigned lds_uniform, int voff) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_uniform), "v"(voff), "s"(rsrc)
               : "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// raw v_exp_f32 (2^x): no denormal-range fix-up; callers pass x <= ~0 where a flushed tiny result is exact enough
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, below fp32 epsilon of the GELU value): 1 rcp + 1 exp + a few FMAs.
// The library erff costs ~40 VALU instructions per element, which made the 128x128 GELU epilogues longer than their
// 12-step main loops.  Returns (erf(|x|/sqrt2) with sign of x, exp(-x^2/2)).
__device__ __forceinline__ float erf_as(float z /* >= 0 */, float e2 /* exp(-z^2) */) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  return 1.0f - p * t * e2;
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float e2 = fast_exp2(-1.4426950408889634f * z * z);
  const float er = copysignf(erf_as(z, e2), x);
  return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float e2 = fast_exp2(-1.4426950408889634f * z * z);  // exp(-x^2/2)
  const float er = copysignf(erf_as(z, e2), x);
  return 0.5f * (1.0f + er) + x * 0.3989422804014327f * e2;
}

// XCD-aware block order for grids whose blocks share an operand stream per (batch, head): the dispatcher deals consecutive block ids round-robin
// over the 8 XCDs (MI355X guide, T1), so with blockIdx.x = tile every XCD's private L2 ends up fetching EVERY head's stream.  With nt tiles per head
// and a 1-D grid of nt * heads * batch blocks, this bijection hands each XCD a contiguous run of (batch, head, tile) triples instead: the blocks that
// are co-resident on one XCD sweep the same head's stream, which its L2 then fetches once.  Speed only - any placement gives the same results.
__device__ __forceinline__ void xcd_block_order(int nt, int heads, int& tile, int& head, int& b) {
  const int n = gridDim.x, id = blockIdx.x;
  const int q8 = n >> 3, r8 = n & 7, xcd = id & 7, slot = id >> 3;
  const int g = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  tile = g % nt;
  head = (g / nt) % heads;
  b = g / (nt * heads);
}

// A/B experiment (MI355X guide, 'Two waves per SIMD' item 9): blocks that share a CU start together and run the same program, i.e. their waves reach
// their matrix chains, their LDS bursts and their VALU blocks in lockstep.  Waves in SIMD wave slot k (HW_ID bits 3:0) sleep k * units * 512 clocks
// once at kernel entry; the block's waves sit in the same slot of four SIMDs, so a block is delayed as a whole.
__device__ __forceinline__ void stagger_by_wave_slot(int units) {
  if (units <= 0) return;
  const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 7;  // HW_REG_HW_ID[3:0]
  for (unsigned i = 0; i < slot * (unsigned)units; i++) __builtin_amdgcn_s_sleep(8);
}

// ---- half sweeps: fp32 accumulators handed from one workgroup to the next (attention_bwd.hip "split sweeps") -------------------------------
struct SweepPart {
  int t0 = 0, t1 = 0;            // sub-range of the swept tiles (t1 = 0: the whole sweep)
  int part = -1;                 // -1 whole sweep, 0 producer half, 1 consumer half
  f32x4* scratch = nullptr;      // this tile's partial accumulators
  unsigned* flag = nullptr;      // 0 -> `want` by the producer, back to 0 by the consumer
  unsigned want = 1u;            // the value that speaks for THIS launch (never 0).  Launches that keep their flags in library-owned device memory stamp it with their
                                 // launch epoch / sequence number: a word left behind by a launch that faulted or never finished is then not mistaken for this one's
  unsigned* status = nullptr;    // the device's sticky fault word (a consumer that gives up waiting)
};
// (the s_nop: a store of more than 8 bytes reads its data registers for two more wait states on gfx940+, and the compiler, which does not look inside the asm,
// is free to put a VALU write of those registers right behind it - it did, in gemm.hip's hand-off: address arithmetic landed in the last piece's first two words)
template <int OFF>
__device__ __forceinline__ void sweep_store16_sc1(f32x4* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1\n\ts_nop 1" ::"v"(p), "v"(v), "i"(OFF) : "memory");
}
// The hand-off in its parts (a kernel whose accumulators fill the register file stores them where they are, in several groups: gemm.hip), and as the two calls the
// attention blocks use.  NV 16-byte pieces per lane and group: piece i of wave w at base[(w * NV + i) * 64 + lane]; a wave's pieces sit 1 KiB apart.
template <int NV>
__device__ __forceinline__ void sweep_store(f32x4* base, const f32x4 (&v)[NV], int w, int l) {
  static_assert(NV % 4 == 0, "pieces go out four to an address register pair");
  f32x4* p = base + (size_t)w * (NV * 64) + l;
#pragma unroll
  for (int i = 0; i < NV; i += 4) {
    sweep_store16_sc1<0>(p + i * 64, v[i]);
    sweep_store16_sc1<1024>(p + i * 64, v[i + 1]);
    sweep_store16_sc1<2048>(p + i * 64, v[i + 2]);
    sweep_store16_sc1<3072>(p + i * 64, v[i + 3]);
  }
}
// every storing wave drains its own stores, the workgroup meets, ONE lane raises the flag that speaks for all of them
__device__ __forceinline__ void sweep_signal(const SweepPart& sp, int t) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) __hip_atomic_store(sp.flag, sp.want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane polls, takes the agent-scope acquire; false (after raising the fault word) if the producer's flag never came.  `ok` lives in LDS the block no longer uses
__device__ __forceinline__ bool sweep_wait(const SweepPart& sp, int t, int* ok) {
  if (t == 0) {
    int spins = 0;
    while (__hip_atomic_load(sp.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sp.want && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(8); spins++; }
    *ok = spins < (1 << 22);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (!*ok) {
    if (t == 0 && sp.status) __hip_atomic_store(sp.status, 0x5004u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return false;
  }
  return true;
}
// v += the partial (ADD) or v = the partial, four loads in flight at a time
template <int NV, bool ADD = true>
__device__ __forceinline__ void sweep_load(const f32x4* base, f32x4 (&v)[NV], int w, int l) {
  const f32x4* p = base + (size_t)w * (NV * 64) + l;
#pragma unroll
  for (int i = 0; i < NV; i += 4) {
    f32x4 a[4];
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = p[(i + j) * 64];
#pragma unroll
    for (int j = 0; j < 4; j++) { if constexpr (ADD) v[i + j] += a[j]; else v[i + j] = a[j]; }
    asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3])::"memory");
  }
}
// every wave holds its addends: the flag goes back to 0 for the next launch
__device__ __forceinline__ void sweep_release(const SweepPart& sp, int t) {
  __syncthreads();
  if (t == 0) __hip_atomic_store(sp.flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NV>
__device__ __forceinline__ void sweep_publish(const SweepPart& sp, const f32x4 (&v)[NV], int t, int w, int l) {
  sweep_store<NV>(sp.scratch, v, w, l);
  sweep_signal(sp, t);
}
template <int NV, bool ADD = true>
__device__ __forceinline__ bool sweep_collect(const SweepPart& sp, f32x4 (&v)[NV], int t, int w, int l, int* ok) {
  if (!sweep_wait(sp, t, ok)) return false;
  sweep_load<NV, ADD>(sp.scratch, v, w, l);
  sweep_release(sp, t);
  return true;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace smbv
