// gemm_nt, second generation: 256 x 128 output tile, 8 waves, LDS ring of 32-deep K slices filled by LDS-DMA, counted vmcnt, one raw
// barrier per slice, epilogue straight from the accumulators with full-line stores.
//
// Why: the 128 x 128 x 64 kernel (gemm.hip) moves 64 FLOP per byte it pulls from L2 into LDS and keeps ONE K-step in flight
// per block - at this model's K = 384..768 it sat at 21 % of the MFMA peak waiting for copies (SQ_WAIT_ANY 0.4-0.5) and spent a third of a
// tile's life in an LDS-staged epilogue.  Here:
//   * 256 x 128 tile: 85 FLOP per staged byte; 8 waves as 4 (M) x 2 (N), each 64 x 64 = 4 x 4 v_mfma_f32_16x16x32_bf16 accumulators;
//   * K in slices of 32 (one MFMA K-step): a slice is [256 + 128 rows][64 B] = 24 KiB = 24 LDS-DMA pieces of 1 KiB, 3 per wave; a ring of
//     NS slots keeps NS-1 slices in flight; each wave waits for ITS pieces with a counted s_waitcnt vmcnt, one s_barrier publishes the
//     slice and retires everybody's reads of the previous one, whose slot the next copy then overwrites (issued from asm, hidden from
//     hipcc's own wait insertion - smbv_common.h); past the end of K the copies run off the buffer and read zeros, so the count stays fixed;
//   * NS = 3 -> 72 KiB of LDS and <= 128 VGPRs: TWO blocks per CU (16 waves), whose main loops and epilogues overlap each other;
//   * 64-byte LDS rows, 16-B chunk c of row r stored at chunk c ^ ((-(r >> 2)) & 3): every ds_read_b128 lane group covers all 16 bank
//     slots; the same involution is applied to the per-lane SOURCE address of the copy (the LDS side of an LDS-DMA is lane-linear);
//   * the B tile is staged with its rows permuted inside each 64-row group (LDS row 16 j + t holds weight row 4 t + j), which costs nothing
//     (a per-lane source address) and makes lane t of the four N-direction MFMAs j = 0..3 own four CONSECUTIVE output columns of a row:
//     every epilogue access (bias, residual / position rows, saved pre-activation, dGELU operand, fp32 and bf16 outputs) is a 16-byte (fp32)
//     or 8-byte (bf16) per-lane piece of a 256 / 128-byte contiguous row segment - no LDS round trip, no wave barriers.
// Reference call sites: the nn.Linear calls of src/models/videomae/modeling_videomae.py:261-264, 311-313, 368-370, 384-386, 720-722, 741 and
// src/models/vjepa/modeling_vjepa.py:251-253, 410-423 (forward and input-gradient products).
#include "smbv_common.h"
#include "smbv_kernels.h"

namespace smbv {
unsigned* device_status_word_dev();

static constexpr int RBM = 256, RBN = 128, RBK = 32;
static constexpr int RSLOT = (RBM + RBN) * RBK * 2;  // 24 KiB
static constexpr int ROPE_LDS_TABLE_FLOATS = 1024;  // EPI 3: the cos | sin table (two floats per thread) in front of the tile's decoded token rows, in the idle ring
// what a thread of an EPI 3 block fetches BEFORE the main loop and parks in registers (the loads' latency is the main loop's): its two floats of the table, and -
// threads 0..255 - one token of the tile decoded into its three table rows.  After the loop they go to LDS between two barriers, no memory access in between
struct RopeRegs { float t0 = 0.f, t1 = 0.f; int p = 0; };   // (p: the token's position id, decoded after the loop - nothing in front of the loop waits for a load)

// Four output rows (one register index r each) of one lane: v[r] are columns n .. n+3 of row m[r].  Loads of all four rows are issued
// before the first use so that they are in flight together; rows past M are clamped for the loads and masked for the stores.
// EPI: 0 plain (bias / row addend / outputs), 1 + GELU (saving the pre-activation), 2 + dGELU multiplier, 3 + rotary embedding of the q | k sections -
// compiled separately to bound registers
template <int EPI>
__device__ __forceinline__ void epi_rows4(const GemmEpi& e, const int (&m)[4], const int M, const int n, f32x4 (&v)[4], const f32x4 bias, const float* rope_lds = nullptr) {
  int mc[4];
  unsigned orow[4];  // 32-bit element offsets throughout (gemm_nt_ring_ok bounds every operand below 2^30 elements): base pointer in SGPRs + one VGPR
#pragma unroll
  for (int r = 0; r < 4; r++) {
    mc[r] = min(m[r], M - 1);
    orow[r] = (unsigned)mc[r];
    v[r] = v[r] * e.alpha + bias;
  }
  if constexpr (EPI != 3) if (e.grp_in > 0) {   // (EPI 3 is a plain bias + bf16 product: gemm_nt refuses the rest)
#pragma unroll
    for (int r = 0; r < 4; r++) orow[r] = (unsigned)((mc[r] / e.grp_in) * e.grp_out + (mc[r] % e.grp_in));
  }
  if constexpr (EPI != 3) if (e.rowadd) {
    unsigned rr[4];
    if (e.rowidx) {
#pragma unroll
      for (int r = 0; r < 4; r++) rr[r] = (unsigned)e.rowidx[mc[r]];
    } else if (e.radd_grp_in > 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) rr[r] = (unsigned)((mc[r] / e.radd_grp_in) * e.radd_grp_out + mc[r] % e.radd_grp_in + e.radd_off);
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) rr[r] = (unsigned)mc[r];
    }
    f32x4 a[4];
#pragma unroll
    for (int r = 0; r < 4; r++) a[r] = *(const f32x4*)(e.rowadd + (rr[r] * (unsigned)e.ldr + (unsigned)n));
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] += a[r];
  }
  if constexpr (EPI == 2) {
    u32x2 p[4];
#pragma unroll
    for (int r = 0; r < 4; r++) p[r] = *(const u32x2*)(e.dgelu_pre + ((unsigned)mc[r] * (unsigned)e.ld_pre + (unsigned)n));
#pragma unroll
    for (int r = 0; r < 4; r++) {
      v[r][0] *= gelu_erf_grad(bf_lo(p[r][0])); v[r][1] *= gelu_erf_grad(bf_hi(p[r][0]));
      v[r][2] *= gelu_erf_grad(bf_lo(p[r][1])); v[r][3] *= gelu_erf_grad(bf_hi(p[r][1]));
    }
  }
  if constexpr (EPI == 1) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (e.aux_pre) {
        const u32x2 p = {pack_bf16(v[r][0], v[r][1]), pack_bf16(v[r][2], v[r][3])};
        if (m[r] < M) *(u32x2*)(e.aux_pre + (orow[r] * (unsigned)e.ld_aux + (unsigned)n)) = p;
        // GELU is applied to the bf16-rounded pre-activation (what autocast's bf16 linear output feeds gelu)
        v[r][0] = bf_lo(p[0]); v[r][1] = bf_hi(p[0]); v[r][2] = bf_lo(p[1]); v[r][3] = bf_hi(p[1]);
      }
      v[r][0] = gelu_erf(v[r][0]); v[r][1] = gelu_erf(v[r][1]); v[r][2] = gelu_erf(v[r][2]); v[r][3] = gelu_erf(v[r][3]);
    }
  }
  if constexpr (EPI == 3) {   // rotary embedding of the q | k sections (GemmEpi::rope_*): on the bf16-rounded product, as the stand-alone kernel read it
    const int seg = e.rope_geo.seg;
    const bool in_section = n < 2 * e.rope_wa;
    const RopeLane rl = rope_lane((n % e.rope_wa) % e.rope_hs, seg, in_section);   // (per lane, once per tile)
    if (rl.on[0] | rl.on[1]) {
      // the tile's 256 tokens were decoded once, by 256 threads, into table-row offsets [3 axes][256 rows] behind the table (ring_epilogue): a lane reads the
      // two axes its pairs turn on - no position load, no decode, in the per-row chain
      const int* trow = (const int*)(rope_lds + ROPE_LDS_TABLE_FLOATS);
      RopeLane rq = rl;
      rq.axis[0] = 0; rq.axis[1] = 1;     // (rope_apply4 selects row[axis]: slots 0 / 1 are filled with what this lane's two pairs need)
      RopeTok tk[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int lr = m[r] & (RBM - 1);   // row inside the tile (tiles start at multiples of RBM)
        tk[r].row[0] = trow[rl.axis[0] * RBM + lr];
        tk[r].row[1] = trow[rl.axis[1] * RBM + lr];
        tk[r].row[2] = 0;
      }
#pragma unroll
      for (int r = 0; r < 4; r++) rope_round_apply4(v[r], rq, tk[r], seg, rope_lds, false);
    }
  }
  if constexpr (EPI != 3) if (e.out_f32) {
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (m[r] < M) *(f32x4*)(e.out_f32 + (orow[r] * (unsigned)e.ld_of + (unsigned)n)) = v[r];
  }
  if (e.out_bf16) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const u32x2 p = {pack_bf16(v[r][0], v[r][1]), pack_bf16(v[r][2], v[r][3])};
      if (m[r] < M) *(u32x2*)(e.out_bf16 + (orow[r] * (unsigned)e.ld_ob + (unsigned)n)) = p;
    }
  }
}

// epilogue shared by the ring kernels: register r of acc[i][j] is row 64 wm + 16 i + 4 (l >> 4) + r, column 64 wn + 4 (l & 15) + j
template <int EPI>
__device__ __forceinline__ void ring_epilogue(const f32x4 (&acc)[4][4], const GemmEpi& e_in, int m0, int n0, int wm, int wn, int l, int M, unsigned char* smem = nullptr,
                                              const RopeRegs rr = RopeRegs()) {
  const GemmEpi e = e_in;
  if constexpr (EPI == 3) {   // the cos | sin table goes into the (now idle) ring: 16 tokens x 8 angles per lane come out of LDS instead of as L1 gathers
    __syncthreads();          // every wave is past its last slice and has drained its run-off copies
    float* ts = (float*)smem;
    ts[threadIdx.x] = rr.t0; ts[512 + threadIdx.x] = rr.t1;
    if (threadIdx.x < RBM) {  // one thread per token of the tile
      int* trow = (int*)smem + ROPE_LDS_TABLE_FLOATS;
      const RopeTok tk = rope_tok(rr.p, e.rope_geo);
      trow[threadIdx.x] = tk.row[0]; trow[RBM + threadIdx.x] = tk.row[1]; trow[2 * RBM + threadIdx.x] = tk.row[2];
    }
    __syncthreads();
  }
  const int n = n0 + 64 * wn + 4 * (l & 15);
  const int mb = m0 + 64 * wm + 4 * (l >> 4);
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (e.bias) bias = *(const f32x4*)(e.bias + n);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int m[4] = {mb + 16 * i, mb + 16 * i + 1, mb + 16 * i + 2, mb + 16 * i + 3};
    f32x4 v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
    epi_rows4<EPI>(e, m, M, n, v, bias, (const float*)smem);
    __builtin_amdgcn_sched_barrier(0);  // keep the next row group's loads from being hoisted up here (register budget of the 2-blocks-per-CU build)
  }
}

// one 16-byte store that writes through to the memory side (sc1): a partial tile another workgroup will read must not sit dirty in this XCD's L2
template <int OFF>   // (the byte offset travels as the instruction's immediate: one address register pair serves four stores)
__device__ __forceinline__ void store16_sc1(f32x4* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1\n\ts_nop 1" ::"v"(p), "v"(v), "i"(OFF) : "memory"); }  // s_nop: see sweep_store16_sc1 (smbv_common.h)
__device__ void splitk_fault(unsigned* status) {
  if (status) __hip_atomic_store(status, 0x6001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// KS = 2 (round 5): the product's K range as two halves per output tile - 2 T workgroups for T tiles, meant for the N = 768 / 384 products with K >= 1,152
// (168 / 240 tiles of 256 x 128: one block on SOME of the 256 CUs), to fill one round of the chip's 512 block slots with tiles of 85 FLOP per staged byte
// instead of 128 x 128 tiles (64).  MEASURED NULL - kept as the op smbv_gemm_nt_splitk with its hand-off test, not selected by the steps
// (profiles/r05_splitk_ab.txt: +0.6 ms per MIM step): 336 half-tile blocks on 256 CUs put two on 80 CUs and one on the rest, so the busiest CU still does
// one whole tile's work - the makespan of the unsplit 168-tile launch - while the mixed 128 x 128 / 128 x 64 launch gives its busiest CU 0.75 of a tile.  Roles are fixed: block b < T computes K half 0 of tile(b) and PUBLISHES its fp32 accumulators (register order, 1 KiB per
// wave-instruction, write-through stores, every wave drained, workgroup barrier, then one flag word); block T + b computes half 1, waits for that flag
// (one lane polls; agent-scope acquire; barrier), adds the partial and runs the epilogue.  Two addends: fp32 addition commutes, there is no ordering to
// fix - run-to-run identical.  A consumer waits only for a producer and producers wait for nothing: no cycle; the producers are the LOWER block ids,
// dispatched first.  The flag is cleared by its consumer; the step zeroes the flag words once per forward (a zero-fill job of the weight-preparation
// launch), a spin that outlasts ~1 s raises the device status word (smbv_device_status 0x6001) and ends.
// ABL: the ablation hooks of tools/gemm_ablate.py (dbg bits: 1 no epilogue traffic, 2 no copies, 4 no MFMAs, 8 no LDS reads) - a build of their own: as tests of
// a run-time argument they sat in the K loop of the product kernel as five taken branches per slice
template <int NS, int EPI, int KS, bool ABL = false>
__global__ __launch_bounds__(512, NS == 3 ? 4 : 2) void gemm_nt_ring_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                                                          int M, int N, int K, int tiles_m, int tiles_n, GemmEpi e_in, int dbg_arg, unsigned* status) {
  const int dbg = ABL ? dbg_arg : 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * RSLOT];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1;
  // bijective XCD remap (blocks b, b+8, ... share an XCD): XCD x owns a contiguous chunk of the tile list, tile_n fastest, so its L2 keeps
  // an A panel while it sweeps the weight matrix
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x, kh = 0;
  if constexpr (KS == 2) { kh = bid >= nwg ? 1 : 0; bid -= kh * nwg; }
  const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
  const int tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
  const int m0 = tile_m * RBM, n0 = tile_n * RBN;
  const int Kh = K / KS;              // this block's K range: [kh * Kh, (kh + 1) * Kh)
  const int nks = Kh / RBK;
  const int k0b = kh * Kh * 2;        // its byte offset inside an operand row

  const i32x4 ra = dma_rsrc(A, (unsigned)(((size_t)(M - 1) * lda + K) * 2));
  const i32x4 rb = dma_rsrc(B, (unsigned)(((size_t)(N - 1) * ldb + K) * 2));
  // copy side: piece = 16 LDS rows x 64 B; lane -> (row l >> 2, physical chunk l & 3) holding logical chunk (l & 3) ^ f, f = (-(row >> 2)) & 3
  const int prow = l >> 2;
  const int csrc = (l & 3) ^ ((-(l >> 4)) & 3);
  const int oa0 = ((m0 + 32 * w + prow) * lda + csrc * 8) * 2 + k0b;       // A rows 32 w .. + 15
  const int oa1 = oa0 + 16 * lda * 2;                                       //        32 w + 16 .. + 31
  const int ob0 = ((n0 + 64 * (w >> 2) + 4 * prow + (w & 3)) * ldb + csrc * 8) * 2 + k0b;  // LDS B row 16 w + prow <- weight row 64 (w >> 2) + 4 prow + (w & 3)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((int)lds_addr_of(smem));
  const unsigned la = lds0 + (unsigned)w * 2048, lb = lds0 + RBM * RBK * 2 + (unsigned)w * 1024;
  auto stage = [&](int s) __attribute__((always_inline)) {
    const unsigned so = (unsigned)(s % NS) * RSLOT;
    const int kb = s * (RBK * 2);
    dma16_hidden(ra, la + so, oa0 + kb);
    dma16_hidden(ra, la + so + 1024, oa1 + kb);
    dma16_hidden(rb, lb + so, ob0 + kb);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // read side: lane -> row (l & 15) of a 16-row fragment tile, logical chunk l >> 4
  const int fr = l & 15;
  const int pch = ((l >> 4) ^ ((-(fr >> 2)) & 3)) * 16;
  const int a_off = (64 * wm + fr) * 64 + pch;
  const int b_off = RBM * RBK * 2 + (64 * wn + fr) * 64 + pch;

  RopeRegs rr;
  if constexpr (EPI == 3) {   // (issued in front of the first copies: older than every piece the counted waits below count)
    const int nf = e_in.rope_geo.rows * e_in.rope_geo.seg;   // <= 1024 (gemm_nt)
    if (t < nf) rr.t0 = e_in.rope_table[t];
    if (t + 512 < nf) rr.t1 = e_in.rope_table[t + 512];
    if (t < RBM) {
      const int m = min(m0 + t, M - 1);
      rr.p = e_in.rope_geo.S > 0 ? rope_row_pos(m, e_in.rope_geo) : 0;   // (computed first: a register a pending load may still write would have to be waited for)
      if (e_in.rope_pos) rr.p = e_in.rope_pos[m];
    }
  }
#pragma unroll
  for (int s = 0; s < NS - 1; s++) stage(s);
  for (int s = 0; s < nks; s++) {
    // own pieces of slice s have landed (the NS-2 younger slices may still fly); the barrier publishes the slice and tells everybody that
    // every wave is done reading slice s-1
    if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!(dbg & 2)) stage(s + NS - 1);  // into the slot of slice s-1 (beyond K: zeros, never read)
    const unsigned char* sp = smem + (s % NS) * RSLOT;
    bf16x8 af[4], bfr[4];
    if (!(dbg & 8)) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        af[i] = *(const bf16x8*)(sp + a_off + i * 1024);
        bfr[i] = *(const bf16x8*)(sp + b_off + i * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) { af[i] = __builtin_bit_cast(bf16x8, (u32x4){(unsigned)s, 1u, 2u, 3u}); bfr[i] = af[i]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are retired before this wave can reach the next barrier
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 4)) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);  // D rows <-> m (4 (l >> 4) + r), cols <-> lane t
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) asm volatile("" ::"v"(af[i]), "v"(bfr[i]));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-off copies still target this block's LDS

  if (dbg & 1) {  // ablation: no epilogue traffic
    if (acc[0][0][0] == 1234.5f) e_in.out_f32[0] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3];
    return;
  }
  if constexpr (KS == 2) {
    unsigned* flag = (unsigned*)e_in.splitk_ws + tile;
    // partial tile: [8 waves][16 accumulators][64 lanes] x 16 B - every wave-instruction moves 1 KiB contiguous, a wave's 16 pieces sit 1 KiB apart
    f32x4* part = (f32x4*)((char*)e_in.splitk_ws + SPLITK_SYNC_BYTES) + (size_t)tile * (RBM * RBN / 4) + w * 1024 + l;
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        f32x4* pi = part + i * 256;
        store16_sc1<0>(pi, acc[i][0]);
        store16_sc1<1024>(pi, acc[i][1]);
        store16_sc1<2048>(pi, acc[i][2]);
        store16_sc1<3072>(pi, acc[i][3]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores ...
      __syncthreads();                                    // ... before the ONE flag store that speaks for all of them
      if (t == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    __shared__ int ok;
    if (t == 0) {
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(8); spins++; }
      ok = spins < (1 << 22);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!ok) { if (t == 0) splitk_fault(status); return; }   // (never on a healthy run: the producer waits for nothing)
#pragma unroll
    for (int i = 0; i < 4; i++) {   // four loads in flight at a time: sixteen would not fit beside the 64 accumulator registers at 128 per lane
      f32x4 pv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) pv[j] = part[(i * 4 + j) * 64];
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] += pv[j];
      // the sums exist HERE (hipcc otherwise keeps all sixteen addends alive and sinks the additions into the epilogue: 40 spilled registers), and the next
      // group's loads stay behind this point
      asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3])::"memory");
    }
    __syncthreads();   // every wave holds its addends (the loads above were waited for by their uses) before the flag goes back to 0 for the next launch
    if (t == 0) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  ring_epilogue<EPI>(acc, e_in, m0, n0, wm, wn, l, M, smem, rr);
}

// The same tile with 64-deep K slices: rows of 128 B, so every copy request is a whole cache line (the 32-deep slices above ask L2 for 64-B
// halves: 4.9 M requests per QKV product against 2.3 M here, profiles/r02_gemm_pmc.txt); 3 slots of 48 KiB = one block per CU, two slices in
// flight, two MFMA K-steps per barrier.  Image swizzle as gemm.hip's swz_nt (16-B chunk c of row r at chunk c ^ ((r >> 1) & 7)).
__device__ __forceinline__ int ring64_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring64_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int M, int N, int K,
                                                                int tiles_m, int tiles_n, GemmEpi e_in) {
  constexpr int SLOT = (RBM + RBN) * 128;  // 48 KiB
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * SLOT];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
  const int tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
  const int m0 = tile_m * RBM, n0 = tile_n * RBN;
  const int nks = K / 64;

  const i32x4 ra = dma_rsrc(A, (unsigned)(((size_t)(M - 1) * lda + K) * 2));
  const i32x4 rb = dma_rsrc(B, (unsigned)(((size_t)(N - 1) * ldb + K) * 2));
  // copy side: piece = 8 LDS rows x 128 B; lane -> (row l >> 3, physical chunk l & 7); pieces at even / odd multiples of 8 rows see
  // (row >> 1) & 7 = (l >> 4) / 4 + (l >> 4): two source-chunk permutations per lane
  const int rin = l >> 3, cpos = l & 7;
  const int cs_e = cpos ^ ((rin >> 1) & 7), cs_o = cpos ^ ((4 + (rin >> 1)) & 7);
  const int oa_e = ((m0 + 32 * w + rin) * lda + cs_e * 8) * 2, oa_o = ((m0 + 32 * w + 8 + rin) * lda + cs_o * 8) * 2;   // A rows 32 w + {0, 8} + rin; +16 rows for pieces 2, 3
  // LDS B row p = 16 w + 8 i + rin (i = 0, 1) <- weight row 64 (w >> 2) + 4 (8 i + rin) + (w & 3)
  const int ob_e = ((n0 + 64 * (w >> 2) + 4 * rin + (w & 3)) * ldb + cs_e * 8) * 2;
  const int ob_o = ((n0 + 64 * (w >> 2) + 4 * (8 + rin) + (w & 3)) * ldb + cs_o * 8) * 2;
  const int a16 = 16 * lda * 2;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((int)lds_addr_of(smem));
  const unsigned la = lds0 + (unsigned)w * 4096, lb = lds0 + RBM * 128 + (unsigned)w * 2048;
  auto stage = [&](int s) __attribute__((always_inline)) {
    const unsigned so = (unsigned)(s % 3) * SLOT;
    const int kb = s * 128;
    dma16_hidden(ra, la + so, oa_e + kb);
    dma16_hidden(ra, la + so + 1024, oa_o + kb);
    dma16_hidden(ra, la + so + 2048, oa_e + a16 + kb);
    dma16_hidden(ra, la + so + 3072, oa_o + a16 + kb);
    dma16_hidden(rb, lb + so, ob_e + kb);
    dma16_hidden(rb, lb + so + 1024, ob_o + kb);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0);
  stage(1);
  for (int s = 0; s < nks; s++) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // own pieces of slice s (slice s + 1 may still fly)
    __builtin_amdgcn_s_barrier();
    stage(s + 2);  // into the slot of slice s - 1 (beyond K: zeros, never read)
    const unsigned char* sa = smem + (s % 3) * SLOT;
    const unsigned char* sb = sa + RBM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 af[4], bfr[4];
      const int ch = (l >> 4) + 4 * ks;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        af[i] = *(const bf16x8*)(sa + ring64_swz(64 * wm + 16 * i + (l & 15), ch));
        bfr[i] = *(const bf16x8*)(sb + ring64_swz(64 * wn + 16 * i + (l & 15), ch));
      }
      if (ks == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all reads of this slice are retired before this wave can reach the next barrier
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ring_epilogue<EPI>(acc, e_in, m0, n0, wm, wn, l, M);
}

bool gemm_nt_ring_ok(int lda, int ldb, int M, int N, int K, const GemmEpi& e) {
  // the epilogue addresses every operand with 32-bit element offsets
  const size_t lim = (size_t)1 << 30;
  const size_t rows_out = e.grp_in > 0 ? (size_t)cdiv(M, e.grp_in) * e.grp_out : (size_t)M;
  const size_t rows_add = e.radd_grp_in > 0 ? (size_t)cdiv(M, e.radd_grp_in) * e.radd_grp_out + e.radd_off : (size_t)M;
  if (rows_out * std::max(std::max(e.ld_ob, e.ld_of), e.ld_aux) >= lim || (size_t)M * e.ld_pre >= lim || (e.rowadd && !e.rowidx && rows_add * e.ldr >= lim) ||
      (e.rowidx && (size_t)e.ldr * (1u << 22) >= ((size_t)1 << 32)))  // indexed rows (position table): at most 4 Mi rows assumed addressable
    return false;
  return M >= 128 && (N % RBN) == 0 && (K % RBK) == 0 && ((size_t)M * lda * 2 < (1ull << 31)) && ((size_t)N * ldb * 2 < (1ull << 31)) &&
         (!e.out_bf16 || ((e.ld_ob & 3) == 0 && ((uintptr_t)e.out_bf16 & 7) == 0)) && (!e.aux_pre || ((e.ld_aux & 3) == 0 && ((uintptr_t)e.aux_pre & 7) == 0)) &&
         (!e.dgelu_pre || ((e.ld_pre & 3) == 0 && ((uintptr_t)e.dgelu_pre & 7) == 0));
}

template <int NS>
static void launch_ring(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e, int dbg, hipStream_t st) {
  const int tm = cdiv(M, RBM), tn = N / RBN;
  const dim3 grid(tm * tn), block(512);
  unsigned* nostatus = nullptr;
  if (dbg && NS == 3 && !e.rope_table) {   // tools/gemm_ablate.py
    if (e.act_gelu) hipLaunchKernelGGL((gemm_nt_ring_kernel<3, 1, 1, true>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e, dbg, nostatus);
    else if (e.dgelu_pre) hipLaunchKernelGGL((gemm_nt_ring_kernel<3, 2, 1, true>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e, dbg, nostatus);
    else hipLaunchKernelGGL((gemm_nt_ring_kernel<3, 0, 1, true>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e, dbg, nostatus);
    return;
  }
  if (e.rope_table) hipLaunchKernelGGL((gemm_nt_ring_kernel<NS, 3, 1>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e, dbg, nostatus);
  else if (e.act_gelu) hipLaunchKernelGGL((gemm_nt_ring_kernel<NS, 1, 1>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e, dbg, nostatus);
  else if (e.dgelu_pre) hipLaunchKernelGGL((gemm_nt_ring_kernel<NS, 2, 1>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e, dbg, nostatus);
  else hipLaunchKernelGGL((gemm_nt_ring_kernel<NS, 0, 1>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e, dbg, nostatus);
}

size_t gemm_nt_splitk_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0 || N % RBN) return 0;
  return SPLITK_SYNC_BYTES + (size_t)cdiv(M, RBM) * (N / RBN) * RBM * RBN * sizeof(float);
}

// What the split-K form takes: 256 x 128 tiles that leave CUs idle (<= 256 tiles) while two halves per tile still fit one round of the 512 block slots, each
// half with enough K steps to amortise the hand-off; plain epilogues only (bias / row addend / remap / outputs).
bool gemm_nt_splitk_wanted(int lda, int ldb, int M, int N, int K, const GemmEpi& e) {
  if (!e.splitk_ws || ((uintptr_t)e.splitk_ws & 255) || e.act_gelu || e.dgelu_pre || e.dot_out) return false;
  if (N % RBN || K % (2 * RBK) || K < 1152 || !gemm_nt_ring_ok(lda, ldb, M, N, K, e)) return false;
  const int tiles = cdiv(M, RBM) * (N / RBN);
  return tiles >= 96 && tiles <= 256 && (size_t)tiles * 4 <= SPLITK_SYNC_BYTES && e.splitk_ws_bytes >= gemm_nt_splitk_workspace_bytes(M, N);
}

int gemm_nt_ring_splitk(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e, hipStream_t st) {
  const int tm = cdiv(M, RBM), tn = N / RBN;
  hipLaunchKernelGGL((gemm_nt_ring_kernel<3, 0, 2>), dim3(2 * tm * tn), dim3(512), 0, st, A, lda, B, ldb, M, N, K, tm, tn, e, 0, device_status_word_dev());
  return check_launch("gemm_nt_ring_splitk");
}

int gemm_nt_ring(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e, int slots, hipStream_t st, int dbg) {
  if (e.act_gelu && e.dgelu_pre) { set_error("gemm_nt_ring: GELU and dGELU epilogues are exclusive"); return SMBV_ERR_ARG; }
  if (slots == 64 && (K % 64) == 0) {
    const int tm = cdiv(M, RBM), tn = N / RBN;
    const dim3 grid(tm * tn), block(512);
    if (e.act_gelu) hipLaunchKernelGGL((gemm_nt_ring64_kernel<1>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e);
    else if (e.dgelu_pre) hipLaunchKernelGGL((gemm_nt_ring64_kernel<2>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e);
    else hipLaunchKernelGGL((gemm_nt_ring64_kernel<0>), grid, block, 0, st, A, lda, B, ldb, M, N, K, tm, tn, e);
    return check_launch("gemm_nt_ring64");
  }
  if (slots == 4) launch_ring<4>(A, lda, B, ldb, M, N, K, e, dbg, st);
  else launch_ring<3>(A, lda, B, ldb, M, N, K, e, dbg, st);
  return check_launch("gemm_nt_ring");
}

}  // namespace smbv
