// MFMA GEMMs for the MIM hot path (gfx950).
//
//   gemm_nt : C[M,N] = epi( A[M,K] * B[N,K]^T )      forward linears (HF weight layout [out,in] is exactly B) and
//                                                    dgrad (with a transposed bf16 weight copy).
//   gemm_tn : C[N1,N2] = sum_m X[m,N1] * Y[m,N2]     wgrad: dW[out,in] = dY[tokens,out]^T * A[tokens,in]; both operands
//                                                    are token-major, the contraction index is the slow one, so the
//                                                    MFMA fragments are fetched with the transposing LDS read
//                                                    (ds_read_b64_tr_b16).  Split over the token range into slabs that
//                                                    a second kernel sums (deterministic: no float atomics).
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 accumulators.
// Global -> registers -> LDS staging, double-buffered LDS, one barrier per K step.  LDS images are XOR-swizzled so
// that every ds_read_b128 / ds_read_b64_tr_b16 lane group hits 16 distinct 16-B slots.
// Reference call sites these replace: src/models/videomae/modeling_videomae.py:261-264, 311-313, 368-370, 384-386,
// 741/801-803, 720-722 (the nn.Linear calls) and their autograd.
#include "smbv_common.h"
#include "smbv_kernels.h"
#include <mutex>

namespace smbv {

static constexpr int BM = 128, BN = 128, BK = 64;
extern int g_variant;  // tuning / A-B hook, see attention_fwd.hip
extern int g_ab;       // SMBV_DEBUG_AB (capi.hip): bit 4 = no token split of the last grouped launch, bit 8 = rotary embedding as launches of its own
unsigned* device_status_word_dev();   // streams.hip

__device__ __forceinline__ int swz_nt(int row, int chunk) { return row * 128 + (((chunk) ^ ((row >> 1) & 7)) << 4); }

// ---------------------------------------------------------------------------------------------------------------
// NT
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                                      int ldb, int M, int N, int K, int tiles_m, GemmEpi e) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * BM * BK * 2];  // [buf][A|B][128*64 bf16] = 64 KiB
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int tile_m = blockIdx.x % tiles_m, tile_n = blockIdx.x / tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = (K + BK - 1) / BK;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int id = t + 256 * i, row = id >> 3, ch = id & 7;
      const int kk = k0 + ch * 8;
      const int gm = m0 + row, gn = n0 + row;
      ra[i] = (gm < M && kk < K) ? *(const u32x4*)(A + (size_t)gm * lda + kk) : (u32x4){0, 0, 0, 0};
      rb[i] = (gn < N && kk < K) ? *(const u32x4*)(B + (size_t)gn * ldb + kk) : (u32x4){0, 0, 0, 0};
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* sa = smem + buf * (2 * BM * BK * 2);
    unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int id = t + 256 * i, row = id >> 3, ch = id & 7;
      *(u32x4*)(sa + swz_nt(row, ch)) = ra[i];
      *(u32x4*)(sb + swz_nt(row, ch)) = rb[i];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const unsigned char* sa = smem + buf * (2 * BM * BK * 2);
    const unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 af[4], bfr[4];
      const int ch = (l >> 4) + 4 * ks;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int ar = wm * 64 + i * 16 + (l & 15);
        const int br = wn * 64 + i * 16 + (l & 15);
        af[i] = *(const bf16x8*)(sa + swz_nt(ar, ch));
        bfr[i] = *(const bf16x8*)(sb + swz_nt(br, ch));
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // D[n][m]
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds m = .. + (l&15), n = .. + 4*(l>>4) + r  (4 consecutive n)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int m = m0 + wm * 64 + i * 16 + (l & 15);
    if (m >= M) continue;
    size_t orow = (size_t)m;
    if (e.grp_in > 0) orow = (size_t)(m / e.grp_in) * e.grp_out + (m % e.grp_in);
    const float* radd = nullptr;
    if (e.rowadd) radd = e.rowadd + (size_t)(e.rowidx ? e.rowidx[m] : (e.radd_grp_in > 0 ? (m / e.radd_grp_in) * e.radd_grp_out + m % e.radd_grp_in + e.radd_off : m)) * e.ldr;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (l >> 4);
      if (n >= N) continue;
      f32x4 v = acc[i][j];
      v *= e.alpha;
      if (e.bias) v += *(const f32x4*)(e.bias + n);
      if (radd) v += *(const f32x4*)(radd + n);
      if (e.dgelu_pre) {
        const u32x2 p = *(const u32x2*)(e.dgelu_pre + (size_t)m * e.ld_pre + n);
        v[0] *= gelu_erf_grad(bf_lo(p[0]));
        v[1] *= gelu_erf_grad(bf_hi(p[0]));
        v[2] *= gelu_erf_grad(bf_lo(p[1]));
        v[3] *= gelu_erf_grad(bf_hi(p[1]));
      }
      if (e.act_gelu) {
        if (e.aux_pre) {
          u32x2 p = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
          *(u32x2*)(e.aux_pre + orow * e.ld_aux + n) = p;
          // GELU is applied to the bf16-rounded pre-activation (what autocast's bf16 linear output feeds gelu)
          v[0] = bf_lo(p[0]); v[1] = bf_hi(p[0]); v[2] = bf_lo(p[1]); v[3] = bf_hi(p[1]);
        }
        v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
      }
      if (e.out_f32) *(f32x4*)(e.out_f32 + orow * e.ld_of + n) = v;
      if (e.out_bf16) {
        u32x2 p = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        *(u32x2*)(e.out_bf16 + orow * e.ld_ob + n) = p;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// NT, LDS-DMA staging (K % 64 == 0): A/B tiles go global -> LDS with buffer_load ... lds (1 KiB per wave-instruction,
// no VGPR round trip, no ds_write); rows beyond M/N come back as zeros from the buffer bounds check.  The LDS image
// is lane-linear per instruction, so the XOR swizzle is applied to the per-lane SOURCE address and again on the read.
// XCD-aware tile order: each XCD gets a contiguous chunk of the (tile_n fastest) tile list, so its L2 keeps a few A
// panels while it sweeps the (small, Infinity-Cache-resident) weight matrix.
// ---------------------------------------------------------------------------------------------------------------
// Epilogue through LDS: each wave parks its 64x64 fp32 accumulator tile (two 32-row halves, 272-B padded rows) in its own 16 KiB
// of the (now idle) staging buffers and re-reads it row-wise, so that every global access of the epilogue - residual / position
// rows, saved pre-activations, fp32 and bf16 outputs - is a fully coalesced 16/32-byte-per-lane row segment instead of
// 8-byte fragments scattered over 16 rows.  (With K = 384..768 the scattered form cost as much as the whole main loop.)
// One 32-row x 64-column piece of a wave's accumulator block (row tiles a0, a1; the accumulators hold D^T: register e of acc[i][j] is
// column j*16 + 4*(lane>>4) + e, row i*16 + (lane&15)), staged through the wave's private LDS area ws and written out row-wise.
__device__ __forceinline__ void epilogue_rows32(float* ws, const f32x4 (&a0)[4], const f32x4 (&a1)[4], const int mbase, const int n, const bool ncol,
                                                const f32x4 b0, const f32x4 b1, const int l, const int M, const GemmEpi& e) {
  constexpr int LDW = 68;
  const int c8 = l & 7, r8 = l >> 3;
  {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      *(f32x4*)(ws + (l & 15) * LDW + j * 16 + 4 * (l >> 4)) = a0[j];
      *(f32x4*)(ws + (16 + (l & 15)) * LDW + j * 16 + 4 * (l >> 4)) = a1[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int r = it * 8 + r8;
      const int m = mbase + r;
      f32x4 v0 = *(const f32x4*)(ws + r * LDW + c8 * 8), v1 = *(const f32x4*)(ws + r * LDW + c8 * 8 + 4);
      float dot = 0.f;
      if (m < M && ncol) {
        size_t orow = (size_t)m;
        if (e.grp_in > 0) orow = (size_t)(m / e.grp_in) * e.grp_out + (m % e.grp_in);
        v0 = v0 * e.alpha + b0;
        v1 = v1 * e.alpha + b1;
        if (e.rowadd) {
          const float* radd = e.rowadd + (size_t)(e.rowidx ? e.rowidx[m] : (e.radd_grp_in > 0 ? (m / e.radd_grp_in) * e.radd_grp_out + m % e.radd_grp_in + e.radd_off : m)) * e.ldr + n;
          v0 += *(const f32x4*)radd;
          v1 += *(const f32x4*)(radd + 4);
        }
        if (e.dgelu_pre) {
          const u32x4 p = *(const u32x4*)(e.dgelu_pre + (size_t)m * e.ld_pre + n);
          v0[0] *= gelu_erf_grad(bf_lo(p[0])); v0[1] *= gelu_erf_grad(bf_hi(p[0]));
          v0[2] *= gelu_erf_grad(bf_lo(p[1])); v0[3] *= gelu_erf_grad(bf_hi(p[1]));
          v1[0] *= gelu_erf_grad(bf_lo(p[2])); v1[1] *= gelu_erf_grad(bf_hi(p[2]));
          v1[2] *= gelu_erf_grad(bf_lo(p[3])); v1[3] *= gelu_erf_grad(bf_hi(p[3]));
        }
        if (e.act_gelu) {
          if (e.aux_pre) {
            u32x4 p = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
            *(u32x4*)(e.aux_pre + orow * e.ld_aux + n) = p;
            // GELU is applied to the bf16-rounded pre-activation (what autocast's bf16 linear output feeds gelu)
            v0[0] = bf_lo(p[0]); v0[1] = bf_hi(p[0]); v0[2] = bf_lo(p[1]); v0[3] = bf_hi(p[1]);
            v1[0] = bf_lo(p[2]); v1[1] = bf_hi(p[2]); v1[2] = bf_lo(p[3]); v1[3] = bf_hi(p[3]);
          }
#pragma unroll
          for (int q = 0; q < 4; q++) { v0[q] = gelu_erf(v0[q]); v1[q] = gelu_erf(v1[q]); }
        }
        if (e.out_f32) {
          *(f32x4*)(e.out_f32 + orow * e.ld_of + n) = v0;
          *(f32x4*)(e.out_f32 + orow * e.ld_of + n + 4) = v1;
        }
        if (e.out_bf16) {
          u32x4 p = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
          *(u32x4*)(e.out_bf16 + orow * e.ld_ob + n) = p;
          if (e.dot_out) {  // this lane's 8 columns of the row-dot (the bf16-rounded outputs, as a reader of the stored matrix would see them)
            const u32x4 o = *(const u32x4*)(e.dot_with + orow * e.ld_dot + n);
            if (e.dot_with_lo) {  // (wave-uniform) dot operand = hi + lo: the fp32 value to 16 significant bits
              const u32x4 ol = *(const u32x4*)(e.dot_with_lo + orow * e.ld_dot + n);
#pragma unroll
              for (int q = 0; q < 4; q++) dot += (bf_lo(o[q]) + bf_lo(ol[q])) * bf_lo(p[q]) + (bf_hi(o[q]) + bf_hi(ol[q])) * bf_hi(p[q]);
            } else {
#pragma unroll
              for (int q = 0; q < 4; q++) dot += bf_lo(o[q]) * bf_lo(p[q]) + bf_hi(o[q]) * bf_hi(p[q]);
            }
          }
        }
      }
      if (e.dot_out) {  // (wave-uniform) the 8 lanes of a row hold its 64 columns: one head of 64 or two heads of 32
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        if (e.dot_dh == 64) dot += __shfl_xor(dot, 4, 64);
        if (m < M && ncol && (c8 & (e.dot_dh / 8 - 1)) == 0)
          e.dot_out[((size_t)(m / e.dot_rows) * (e.dot_heads) + n / e.dot_dh) * e.dot_rows + m % e.dot_rows] = dot;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

__device__ __forceinline__ void epilogue_nt_lds(f32x4 (&acc)[4][4], unsigned char* smem, int w, int m0, int n0, int wm, int wn, int l, int M,
                                                int N, const GemmEpi& e_in) {
  const GemmEpi e = e_in;  // by-value copy: keeps the kernel-argument struct from being pinned in memory
  float* ws = (float*)(smem + w * 16384);
  const int n = n0 + wn * 64 + (l & 7) * 8;
  const bool ncol = n < N;
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (e.bias && ncol) { b0 = *(const f32x4*)(e.bias + n); b1 = *(const f32x4*)(e.bias + n + 4); }
  epilogue_rows32(ws, acc[0], acc[1], m0 + wm * 64, n, ncol, b0, b1, l, M, e);
  epilogue_rows32(ws, acc[2], acc[3], m0 + wm * 64 + 32, n, ncol, b0, b1, l, M, e);
}

// One output tile of the LDS-DMA NT kernel.  HALF = false: 128 x 128, waves 2 x 2 of 64 x 64.  HALF = true: 128 x 64 (the mixed-tile launch below),
// waves 4 x 1 of 32 x 64 - the same MFMA per output element in the same K order, so a product is bitwise independent of how its tiles were cut.
template <bool HALF>
__device__ __forceinline__ void nt_dma_tile(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int M, int N, int K, int m0, int n0,
                                            const GemmEpi& e, unsigned char* smem, int dbg) {
  constexpr int NI = HALF ? 2 : 4;  // 16-row fragments of A per wave
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = HALF ? w : (w >> 1), wn = HALF ? 0 : (w & 1);
  const int nk = K / BK;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((size_t)(M - 1) * lda + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((size_t)(N - 1) * ldb + K) * 2), 0x00020000);
  // this lane's source offsets for the 4 pieces (8 rows each) its wave stages per operand
  // piece i (i = 0..3) of this wave covers tile rows (4w + i)*8 .. +7; lane -> (row_in = l>>3, chunk position = l&7).
  // (row >> 1) & 7 == (4*(i&1) + (row_in >> 1)) & 7, so only two distinct source-chunk permutations exist per lane.
  const int rin = l >> 3, cpos = l & 7;
  const int cs_even = cpos ^ ((rin >> 1) & 7), cs_odd = cpos ^ ((4 + (rin >> 1)) & 7);
  const int row_base = w * 32 + rin;
  const int oa_e = ((m0 + row_base) * lda + cs_even * 8) * 2, oa_o = ((m0 + row_base + 8) * lda + cs_odd * 8) * 2;
  const int ob_e = ((n0 + row_base) * ldb + cs_even * 8) * 2, ob_o = ((n0 + row_base + 8) * ldb + cs_odd * 8) * 2;
  const int sa16 = lda * 32, sb16 = ldb * 32;  // 16 rows further, in bytes
  const bool stage_b = !HALF || w < 2;         // a half tile has 64 weight rows: waves 0 and 1 copy them
  auto stage = [=](int buf, int kt) __attribute__((always_inline)) {
    unsigned char* sa = smem + buf * (2 * BM * BK * 2) + w * 4096;
    unsigned char* sb = sa + BM * BK * 2;
    const int kb = kt * BK * 2;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sa), 16, oa_e + kb, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sa + 1024), 16, oa_o + kb, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sa + 2048), 16, oa_e + sa16 + kb, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sa + 3072), 16, oa_o + sa16 + kb, 0, 0, 0);
    if (stage_b) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(sb), 16, ob_e + kb, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(sb + 1024), 16, ob_o + kb, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(sb + 2048), 16, ob_e + sb16 + kb, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(sb + 3072), 16, ob_o + sb16 + kb, 0, 0, 0);
    }
  };

  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  __syncthreads();  // (hipcc drains the pending LDS-DMA with vmcnt(0) in front of the barrier)
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
    const unsigned char* sa = smem + buf * (2 * BM * BK * 2);
    const unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 af[NI], bfr[4];
      const int ch = (l >> 4) + 4 * ks;
#pragma unroll
      for (int i = 0; i < NI; i++) af[i] = *(const bf16x8*)(sa + swz_nt(wm * (16 * NI) + i * 16 + (l & 15), ch));
#pragma unroll
      for (int j = 0; j < 4; j++) bfr[j] = *(const bf16x8*)(sb + swz_nt(wn * 64 + j * 16 + (l & 15), ch));
#pragma unroll
      for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // D[n][m]
    }
    __syncthreads();
  }
  if (dbg == 1) { if (acc[0][0][0] == 1234.5f) e.out_f32[0] = acc[1][1][1] + acc[1][2][2] + acc[1][3][3]; return; }
  if constexpr (!HALF) {
    epilogue_nt_lds(acc, smem, w, m0, n0, wm, wn, l, M, N, e);
  } else {
    const GemmEpi ee = e;  // by-value copy: keeps the kernel-argument struct from being pinned in memory
    float* ws = (float*)(smem + w * 16384);
    const int n = n0 + (l & 7) * 8;
    const bool ncol = n < N;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (ee.bias && ncol) { b0 = *(const f32x4*)(ee.bias + n); b1 = *(const f32x4*)(ee.bias + n + 4); }
    epilogue_rows32(ws, acc[0], acc[1], m0 + w * 32, n, ncol, b0, b1, l, M, ee);
  }
}

// full_rows = tiles_m: every tile is 128 x 128.  full_rows < tiles_m (the mixed-tile launch): tile rows full_rows .. tiles_m - 1 are cut into 128 x 64
// half tiles, 2 tiles_n per row.  A grid of 256 < T <= 384 full tiles puts two tiles on some CUs and one on the others (all blocks are co-resident, two per
// CU); cutting the last rows so that the launch has <= 512 blocks, the halves issued AFTER the full tiles of the same XCD, pairs every full tile with a
// half: 1.5 tiles of work on the busiest CU instead of 2 (the encoder's N = 768 products: 336 tiles -> 168 + 336 halves).
__global__ __launch_bounds__(256, 2) void gemm_nt_dma_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                                             int M, int N, int K, int tiles_m, int tiles_n, int full_rows, GemmEpi e, int dbg) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * BM * BK * 2];  // [buf][A|B][128 rows][128 B]
  // bijective XCD remap (blocks b, b+8, ... share an XCD): XCD x owns a contiguous chunk of the tile list - of each kind of tile
  const int bid = blockIdx.x, xcd = bid & 7, pos = bid >> 3;
  const int n_full = full_rows * tiles_n, n_half = (tiles_m - full_rows) * tiles_n * 2;
  auto chunk_tile = [&](int n, int p) { const int qd = n >> 3, rm = n & 7; return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + p; };
  const int mine_full = (n_full >> 3) + (xcd < (n_full & 7) ? 1 : 0);
  if (pos < mine_full) {
    const int tile = chunk_tile(n_full, pos);
    nt_dma_tile<false>(A, lda, B, ldb, M, N, K, (tile / tiles_n) * BM, (tile % tiles_n) * BN, e, smem, dbg);
  } else {
    const int tile = chunk_tile(n_half, pos - mine_full);
    nt_dma_tile<true>(A, lda, B, ldb, M, N, K, (full_rows + tile / (2 * tiles_n)) * BM, (tile % (2 * tiles_n)) * (BN / 2), e, smem, dbg);
  }
}

// the row-dot by-product lives in the LDS-staged epilogue of the LDS-DMA kernel (a wave's 64 output columns = whole heads)
static bool nt_dma_ok(int lda, int ldb, int M, int N, int K, const GemmEpi& e) {
  return (K % BK == 0) && (N % 8 == 0) && (!e.out_bf16 || ((e.ld_ob & 7) == 0 && ((uintptr_t)e.out_bf16 & 15) == 0)) &&
         (!e.aux_pre || ((e.ld_aux & 7) == 0 && ((uintptr_t)e.aux_pre & 15) == 0)) &&
         (!e.dgelu_pre || ((e.ld_pre & 7) == 0 && ((uintptr_t)e.dgelu_pre & 15) == 0)) && ((size_t)M * lda * 2 < (1ull << 31)) && ((size_t)N * ldb * 2 < (1ull << 31));
}
bool gemm_nt_rowdot_ok(int lda, int ldb, int M, int N, int K, const GemmEpi& e) {
  const bool ring_shape = (K <= 768 && N >= 1024) || (K <= 1024 && N >= 2048);
  const bool ring = gemm_nt_ring_ok(lda, ldb, M, N, K, e) && !(g_variant & 256) && (ring_shape || (g_variant & 131072));
  return !ring && !(g_variant & 32) && nt_dma_ok(lda, ldb, M, N, K, e) && e.out_bf16 && (N % 64) == 0 && (e.ld_dot & 7) == 0;
}

int gemm_nt(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0) { set_error("gemm_nt: empty problem M=%d N=%d K=%d", M, N, K); return SMBV_ERR_SHAPE; }
  if ((K & 7) || (lda & 7) || (ldb & 7) || (N & 3)) { set_error("gemm_nt: K,lda,ldb must be multiples of 8 and N of 4 (K=%d lda=%d ldb=%d N=%d)", K, lda, ldb, N); return SMBV_ERR_SHAPE; }
  if (((uintptr_t)A | (uintptr_t)B) & 15) { set_error("gemm_nt: operands must be 16-byte aligned"); return SMBV_ERR_ALIGN; }
  if ((e.out_bf16 && ((e.ld_ob & 3) || ((uintptr_t)e.out_bf16 & 7))) || (e.out_f32 && ((e.ld_of & 3) || ((uintptr_t)e.out_f32 & 15))) ||
      (e.bias && ((uintptr_t)e.bias & 15)) || (e.rowadd && ((e.ldr & 3) || ((uintptr_t)e.rowadd & 15))) ||
      (e.dgelu_pre && ((e.ld_pre & 3) || ((uintptr_t)e.dgelu_pre & 7))) || (e.aux_pre && ((e.ld_aux & 3) || ((uintptr_t)e.aux_pre & 7)))) {
    set_error("gemm_nt: epilogue operand alignment/stride"); return SMBV_ERR_ALIGN;
  }
  const bool ring_shape = (K <= 768 && N >= 1024) || (K <= 1024 && N >= 2048);
  if (e.rope_table) {   // rotary embedding of the q | k sections on the way out: the ring kernel's epilogue, or the product followed by the stand-alone launch
    const int seg = 2 * ((e.rope_Dh / 3) / 2);
    if (!e.out_bf16 || e.out_f32 || e.act_gelu || e.dgelu_pre || e.rowadd || e.grp_in > 0 || e.dot_out || e.alpha != 1.0f || e.rope_wa <= 0 || e.rope_hs <= 0 ||
        (e.rope_wa & 7) || (e.rope_hs & 7) || e.rope_wa % e.rope_hs || 2 * e.rope_wa > N || e.rope_grid <= 0 || (!e.rope_pos && e.rope_S <= 0) || 3 * seg > e.rope_hs) {
      set_error("gemm_nt: rotary epilogue needs a plain bf16 product with q | k sections of whole head slots (wa=%d hs=%d Dh=%d N=%d)", e.rope_wa, e.rope_hs, e.rope_Dh, N);
      return SMBV_ERR_ARG;
    }
    RopeGeo geo;
    const bool fused = seg > 0 && rope_geo(geo, e.rope_grid, e.rope_Dh, e.rope_rows) && e.rope_rows * seg <= 1024 /* two table floats per thread */ &&
                       (e.rope_pos || rope_geo_rows(geo, M, e.rope_S)) && ring_shape && gemm_nt_ring_ok(lda, ldb, M, N, K, e) &&
                       !(g_variant & (256 | 262144 | 512)) && !(g_ab & 8);
    if (fused) {
      GemmEpi ee = e;
      ee.rope_geo = geo;
      return gemm_nt_ring(A, lda, B, ldb, M, N, K, ee, 3, st, (g_variant >> 20) & 15);
    }
    if (!fused) {
      GemmEpi plain = e;
      plain.rope_table = nullptr;
      const int rc = gemm_nt(A, lda, B, ldb, M, N, K, plain, st);
      if (rc != SMBV_OK || seg <= 0) return rc;
      return rope3d(e.out_bf16, e.ld_ob, e.rope_wa, M, e.rope_S, e.rope_wa / e.rope_hs, e.rope_hs, e.rope_Dh, e.rope_grid, e.rope_pos, e.rope_table, 0, st);
    }
  }
  // long-K products whose 256 x 128 tiles leave CUs idle: two K halves per tile (gemm_nt_ring.hip, round 5) when the caller handed over split-K scratch
  if (gemm_nt_splitk_wanted(lda, ldb, M, N, K, e) && !(g_variant & (256 | 131072))) return gemm_nt_ring_splitk(A, lda, B, ldb, M, N, K, e, st);
  // Kernel choice, measured per shape on MI355X (tools/gemm_ab.py, profiles/r02_gemm_ab.txt): the ring kernel (256 x 128 tiles, two K slices in
  // flight per block, register epilogue) is 7-19 % ahead where the epilogue is heavy and K short - the GELU / dGELU products, the decoder's
  // K = 384 projections, the head - and 5-13 % behind the 128 x 128 kernel on long-K or narrow outputs, whose smaller tiles fill the chip better.
  if (gemm_nt_ring_ok(lda, ldb, M, N, K, e) && !(g_variant & 256) && (ring_shape || (g_variant & 131072)))
    return gemm_nt_ring(A, lda, B, ldb, M, N, K, e, (g_variant & 262144) ? 64 : ((g_variant & 512) ? 4 : 3), st, (g_variant >> 20) & 15);
  if (e.dot_out && (!e.dot_with || (e.dot_dh != 64 && e.dot_dh != 32) || e.dot_rows <= 0 || (((uintptr_t)e.dot_with | (uintptr_t)e.dot_with_lo) & 15) || !gemm_nt_rowdot_ok(lda, ldb, M, N, K, e))) {
    set_error("gemm_nt: row-dot by-product not available for this product (check gemm_nt_rowdot_ok first)"); return SMBV_ERR_ARG;
  }
  const int tm = cdiv(M, BM), tn = cdiv(N, BN);
  const bool dma_ok = nt_dma_ok(lda, ldb, M, N, K, e);
  if (dma_ok && !(g_variant & 32)) {
    // mixed-tile launch (see the kernel): grids of 256 < T <= 384 tiles whose N is a whole number of tiles; the full rows are what 512 block slots leave
    int full_rows = tm;
    const int T = tm * tn;
    if (T > 256 && T <= 384 && N % BN == 0 && !(g_variant & 1024)) {
      const int need = 2 * T - 512;                       // full tiles so that full + halves <= 512 blocks
      int fr = std::max(0, cdiv(need, tn));
      // both kinds split evenly over the 8 XCDs (the kernel hands each XCD a contiguous chunk of each kind, full tiles first)
      while (fr < tm && (((fr * tn) & 7) || (((tm - fr) * tn * 2) & 7))) fr++;
      if (fr < tm) full_rows = fr;
    }
    const int blocks = full_rows * tn + (tm - full_rows) * tn * 2;
    GemmEpi ee = e;
    ee.dot_heads = e.dot_out ? N / e.dot_dh : 0;
    hipLaunchKernelGGL(gemm_nt_dma_kernel, dim3(blocks), dim3(256), 0, st, A, lda, B, ldb, M, N, K, tm, tn, full_rows, ee, (g_variant & 64) ? 1 : ((g_variant & 128) ? 2 : 0));
    return check_launch("gemm_nt_dma");
  }
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(tm * tn), dim3(256), 0, st, A, lda, B, ldb, M, N, K, tm, e);
  return check_launch("gemm_nt");
}

// ---------------------------------------------------------------------------------------------------------------
// TN (wgrad)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sw_tn(int m) { return (m & 3) | (((m >> 3) & 1) << 2); }

// Bias gradient from the same pass: the column sums of X over the tokens are one more product, X^T * ones.  Blocks of the first N2 tile
// accumulate it with two extra MFMAs per wave and K-step (an all-ones A operand: every row of that accumulator tile holds the sums) on the
// X fragments they already hold - the separate column-sum passes over dY (two launches and a full re-read per linear) are gone.
__device__ __forceinline__ bf16x8 ones_frag() {
  const u32x4 o = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  return __builtin_bit_cast(bf16x8, o);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ Y,
                                                      int ldy, int M, int N1, int N2, float* __restrict__ out, int ldo,
                                                      int tiles_1, int mchunk, size_t slab_stride, float* __restrict__ xsum) {
  // LDS: [buf][X|Y][64 m][128 n] bf16 (256-B rows) = 2*2*16 KiB
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * BK * 128 * 2];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int w1 = w >> 1, w2 = w & 1;  // wave's 64-wide slice of n1 / n2
  const int tile_1 = blockIdx.x % tiles_1, tile_2 = blockIdx.x / tiles_1;
  const int a0 = tile_1 * 128, b0 = tile_2 * 128;  // n1 / n2 origin
  const int z = blockIdx.y;
  const int mbeg = z * mchunk, mend = min(M, mbeg + mchunk);
  const int nk = (mend - mbeg + BK - 1) / BK;

  f32x4 acc[4][4];  // [n1 tile i][n2 tile j], D rows <-> n2, cols <-> n1
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_sum = xsum != nullptr && tile_2 == 0;
  f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

  u32x4 rx[4], ry[4];
  auto gload = [&](int kt) {
    const int mm0 = mbeg + kt * BK;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int id = t + 256 * i, row = id >> 4, c16 = id & 15;
      const int gm = mm0 + row;
      const int ga = a0 + c16 * 8, gb = b0 + c16 * 8;
      rx[i] = (gm < mend && ga < N1) ? *(const u32x4*)(X + (size_t)gm * ldx + ga) : (u32x4){0, 0, 0, 0};
      ry[i] = (gm < mend && gb < N2) ? *(const u32x4*)(Y + (size_t)gm * ldy + gb) : (u32x4){0, 0, 0, 0};
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* sx = smem + buf * (2 * BK * 256);
    unsigned char* sy = sx + BK * 256;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int id = t + 256 * i, row = id >> 4, c16 = id & 15;
      const int off = row * 256 + 32 * ((c16 >> 1) ^ sw_tn(row)) + 16 * (c16 & 1);
      *(u32x4*)(sx + off) = rx[i];
      *(u32x4*)(sy + off) = ry[i];
    }
  };

  if (nk > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  const int g = l >> 4, q = (l >> 2) & 3, p = l & 3;
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const unsigned char* sx = smem + buf * (2 * BK * 256);
    const unsigned char* sy = sx + BK * 256;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 xf[4], yf[4];
      const int mA = 32 * ks + 8 * g + q, mB = mA + 4;
      const int swA = sw_tn(mA), swB = sw_tn(mB);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int nt1 = w1 * 4 + i, nt2 = w2 * 4 + i;
        s16x4 x0 = lds_read_tr16(sx + mA * 256 + 32 * (nt1 ^ swA) + 8 * p);
        s16x4 x1 = lds_read_tr16(sx + mB * 256 + 32 * (nt1 ^ swB) + 8 * p);
        s16x4 y0 = lds_read_tr16(sy + mA * 256 + 32 * (nt2 ^ swA) + 8 * p);
        s16x4 y1 = lds_read_tr16(sy + mB * 256 + 32 * (nt2 ^ swB) + 8 * p);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 xx = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        s16x8 yy = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
        xf[i] = __builtin_bit_cast(bf16x8, xx);
        yf[i] = __builtin_bit_cast(bf16x8, yy);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[j], xf[i], acc[i][j], 0, 0, 0);  // D[n2][n1]
      if (do_sum) {
        accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_frag(), w2 ? xf[2] : xf[0], accb[0], 0, 0, 0);
        accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_frag(), w2 ? xf[3] : xf[1], accb[1], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  if (do_sum && l < 16) {  // wave (w1, w2) owns column tiles 2 w2, 2 w2 + 1 of its n1 slice
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int n1 = a0 + w1 * 64 + (2 * w2 + k) * 16 + l;
      if (n1 < N1) xsum[(size_t)z * N1 + n1] = accb[k][0];
    }
  }
  float* o = out + (size_t)z * slab_stride;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int n1 = a0 + w1 * 64 + i * 16 + (l & 15);
    if (n1 >= N1) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n2 = b0 + w2 * 64 + j * 16 + 4 * (l >> 4);
      if (n2 >= N2) continue;
      *(f32x4*)(o + (size_t)n1 * ldo + n2) = acc[i][j];
    }
  }
}

// TN with LDS-DMA staging (N1, N2 multiples of 128): 4 rows x 256 B per wave-instruction; the 32-byte-column XOR swizzle is
// applied to the per-lane source address; rows beyond this split's token range read as zeros through the buffer bounds check.
// (Round 3: the same kernel with 32-token slices through a four-slot ring - three slices in flight behind a counted vmcnt, one barrier per slice -
// measured 0..12 % SLOWER at every shape of the step, profiles/r03_gemm_tn_ring_ab.txt: with two blocks per CU the other block's MFMAs already cover
// a step's copy, and the ring doubles the barriers.  Not kept.)
// One 128 x 128 output tile over the token range [mbeg, mend): out tile rows a0.., columns b0.. (leading dimension ldo); xsum != null: this block also
// forms the column sums of X over its token range for its 128 columns (do_sum), delivered by tn_store_sum.
struct TnSum { float *d0, *d1, *d2; int seg; };  // where column sum n of X goes: d{n / seg}[n % seg] (null = dropped); seg = 0: xsum_flat[n]
__device__ __forceinline__ void tn_store_sum(const TnSum& ts, float* xsum_flat, int n, float v) {
  if (ts.seg > 0) {
    const int sg = n / ts.seg;
    float* d = sg == 0 ? ts.d0 : (sg == 1 ? ts.d1 : ts.d2);   // (selects, not an indexed array: hipcc would promote that to LDS)
    if (d) d[n - sg * ts.seg] = v;
  } else {
    xsum_flat[n] = v;
  }
}
__device__ __forceinline__ void tn128_tile(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ Y, int ldy, int N1, int N2, float* __restrict__ o,
                                           int ldo, int a0, int b0, int mbeg, int mend, bool do_sum, float* __restrict__ xsum_flat, const TnSum& ts,
                                           unsigned char* smem) {
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int w1 = w >> 1, w2 = w & 1;
  const int nk = (mend - mbeg + BK - 1) / BK;

  // (the copies are issued from asm statements - smbv_common.h - because hipcc puts a vmcnt(0) in front of the first ds_read_b64_tr_b16
  // that follows an LDS-DMA it can see, which serialised every K-step's copy with its MFMAs)
  const i32x4 rx = dma_rsrc(X, mend > 0 ? (unsigned)(((size_t)(mend - 1) * ldx + N1) * 2) : 0u);
  const i32x4 ry = dma_rsrc(Y, mend > 0 ? (unsigned)(((size_t)(mend - 1) * ldy + N2) * 2) : 0u);
  // piece i (0..3) of wave w covers tile rows 16w + 4i .. +3; lane -> (row_in = l>>4, 16-B chunk position = l&15)
  const int rin = l >> 4, cp = l & 15;
  int ox[4], oy[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = 16 * w + 4 * i + rin;
    const int sw = (rin & 3) | (((i >> 1) & 1) << 2);            // == sw_tn(row)
    const int csrc = 2 * ((cp >> 1) ^ sw) + (cp & 1);
    ox[i] = ((mbeg + row) * ldx + a0 + csrc * 8) * 2;
    oy[i] = ((mbeg + row) * ldy + b0 + csrc * 8) * 2;
  }
  const int ox0 = ox[0], ox1 = ox[1], ox2 = ox[2], ox3 = ox[3], oy0 = oy[0], oy1 = oy[1], oy2 = oy[2], oy3 = oy[3];
  const int kx = BK * ldx * 2, ky = BK * ldy * 2;
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem) + w * 4096));
  auto stage = [=](int buf, int kt) __attribute__((always_inline)) {
    const unsigned sx = lds_w + (unsigned)buf * (2 * BK * 256), sy = sx + BK * 256;
    dma16_hidden(rx, sx, ox0 + kt * kx);
    dma16_hidden(ry, sy, oy0 + kt * ky);
    dma16_hidden(rx, sx + 1024, ox1 + kt * kx);
    dma16_hidden(ry, sy + 1024, oy1 + kt * ky);
    dma16_hidden(rx, sx + 2048, ox2 + kt * kx);
    dma16_hidden(ry, sy + 2048, oy2 + kt * ky);
    dma16_hidden(rx, sx + 3072, ox3 + kt * kx);
    dma16_hidden(ry, sy + 3072, oy3 + kt * ky);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

  if (nk > 0) stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int g = l >> 4, q = (l >> 2) & 3, p = l & 3;
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
    const unsigned char* sx = smem + buf * (2 * BK * 256);
    const unsigned char* sy = sx + BK * 256;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 xf[4], yf[4];
      const int mA = 32 * ks + 8 * g + q, mB = mA + 4;
      const int swA = sw_tn(mA), swB = sw_tn(mB);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int nt1 = w1 * 4 + i, nt2 = w2 * 4 + i;
        s16x4 x0 = lds_read_tr16(sx + mA * 256 + 32 * (nt1 ^ swA) + 8 * p);
        s16x4 x1 = lds_read_tr16(sx + mB * 256 + 32 * (nt1 ^ swB) + 8 * p);
        s16x4 y0 = lds_read_tr16(sy + mA * 256 + 32 * (nt2 ^ swA) + 8 * p);
        s16x4 y1 = lds_read_tr16(sy + mB * 256 + 32 * (nt2 ^ swB) + 8 * p);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 xx = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        s16x8 yy = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
        xf[i] = __builtin_bit_cast(bf16x8, xx);
        yf[i] = __builtin_bit_cast(bf16x8, yy);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[j], xf[i], acc[i][j], 0, 0, 0);  // D[n2][n1]
      if (do_sum) {
        accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_frag(), w2 ? xf[2] : xf[0], accb[0], 0, 0, 0);
        accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_frag(), w2 ? xf[3] : xf[1], accb[1], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next K-step's copy had this step's MFMAs to land
    __syncthreads();
  }

  if (do_sum && l < 16) {
#pragma unroll
    for (int k = 0; k < 2; k++) tn_store_sum(ts, xsum_flat, a0 + w1 * 64 + (2 * w2 + k) * 16 + l, accb[k][0]);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int n1 = a0 + w1 * 64 + i * 16 + (l & 15);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n2 = b0 + w2 * 64 + j * 16 + 4 * (l >> 4);
      *(f32x4*)(o + (size_t)n1 * ldo + n2) = acc[i][j];
    }
  }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_dma_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ Y, int ldy,
                                                             int M, int N1, int N2, float* __restrict__ out, int ldo, int tiles_1,
                                                             int mchunk, size_t slab_stride, float* __restrict__ xsum) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * BK * 128 * 2];
  const int tile_1 = blockIdx.x % tiles_1, tile_2 = blockIdx.x / tiles_1;
  const int z = blockIdx.y;
  const int mbeg = z * mchunk, mend = min(M, mbeg + mchunk);
  const TnSum none = {nullptr, nullptr, nullptr, 0};
  tn128_tile(X, ldx, Y, ldy, N1, N2, out + (size_t)z * slab_stride, ldo, tile_1 * 128, tile_2 * 128, mbeg, mend, xsum != nullptr && tile_2 == 0,
             xsum ? xsum + (size_t)z * N1 : nullptr, none, smem);
}

// TN, 256 x 256 output tile, for wide outputs (gemm_tn_use_256).  A 128 x 128 block is bound by its own copy latency: its LDS (64 KiB,
// 2 blocks per CU) lets 2 x 32 KiB fly per CU.  A 256 x 256 tile needs half the operand bytes per MAC and, alone on its CU, keeps three
// 32-token slices (96 KiB) in flight: 8 waves (2 x 4), each 128 x 64 of the output, slices [32 tokens][512 B] of X and of Y
// through a ring of four LDS slots filled by LDS-DMA hidden from hipcc (smbv_common.h), one counted vmcnt + one barrier per slice, fragments
// by ds_read_b64_tr_b16 (same 32-byte-column XOR swizzle as the 128 kernel; 512-B rows keep its bank pattern).  The token range is split
// into slabs so that tiles x slabs fills one round of 256 blocks; slabs are summed by reduce_tn_kernel in slab order.
template <bool SPLIT = false>   // SPLIT: the tile may be one half of a token range (sp.part >= 0) - a separate build, so that the whole-range kernels keep their registers
__device__ __forceinline__ void tn256_tile(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ Y, int ldy, int N1, int N2, float* __restrict__ o,
                                           int ldo, int a0, int b0, int mbeg, int mend, bool do_sum, float* __restrict__ xsum_flat, const TnSum& ts,
                                           unsigned char* smem, const SweepPart sp = SweepPart()) {
  constexpr int SLICE = 2 * 32 * 512;  // X image | Y image
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int w1 = w >> 2, w2 = w & 3;  // 2 (n1) x 4 (n2)
  const int nks = (mend - mbeg + 31) / 32;

  const i32x4 rx = dma_rsrc(X, mend > 0 ? (unsigned)(((size_t)(mend - 1) * ldx + N1) * 2) : 0u);
  const i32x4 ry = dma_rsrc(Y, mend > 0 ? (unsigned)(((size_t)(mend - 1) * ldy + N2) * 2) : 0u);
  // wave w stages token rows 4w..4w+3 of both images as 2 + 2 pieces of 2 rows x 512 B (lane -> row l>>5, 16-B chunk position l&31)
  const int cp = l & 31;
  int ox[2], oy[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = 4 * w + 2 * i + (l >> 5);
    const int csrc = 2 * ((cp >> 1) ^ sw_tn(row)) + (cp & 1);
    ox[i] = ((mbeg + row) * ldx + a0 + csrc * 8) * 2;
    oy[i] = ((mbeg + row) * ldy + b0 + csrc * 8) * 2;
  }
  const int ox0 = ox[0], ox1 = ox[1], oy0 = oy[0], oy1 = oy[1];
  const int kx = 32 * ldx * 2, ky = 32 * ldy * 2;
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem) + w * 2048));
  auto stage = [&](int ks) __attribute__((always_inline)) {
    const unsigned dx = lds_w + (unsigned)(ks & 3) * SLICE, dy = dx + 32 * 512;
    dma16_hidden(rx, dx, ox0 + ks * kx);
    dma16_hidden(ry, dy, oy0 + ks * ky);
    dma16_hidden(rx, dx + 1024, ox1 + ks * kx);
    dma16_hidden(ry, dy + 1024, oy1 + ks * ky);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // (do_sum) wave (w1, w2) sums column tiles 2 w2, 2 w2 + 1 of its 128-wide n1 slice
  f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

  const int g = l >> 4, q = (l >> 2) & 3, p = l & 3;
  const int mA = 8 * g + q, mB = mA + 4;
  const int swA = sw_tn(mA), swB = sw_tn(mB);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // Ping-pong: the block's two wave groups (w1 = 0 / 1: one wave of each per SIMD) run half a slice apart - while one group reads its
  // fragments of a slice from LDS, the other issues its 32 MFMAs, so every SIMD always has a wave feeding the matrix pipe.  Both groups
  // execute the same barrier sequence (two per slice); group 1 is shifted by one barrier (an extra one in front, group 0 makes it up
  // at the end).  Barrier 2k is "slice k is readable" for group 0, barrier 2k+1 for group 1; every wave has waited for its own pieces
  // of slice k (vmcnt) before barrier 2k, whichever phase it is in.
  auto read_frags = [&](int ks, bf16x8* xf, bf16x8* yf) __attribute__((always_inline)) {
    const unsigned char* sx = smem + (ks & 3) * SLICE;
    const unsigned char* sy = sx + 32 * 512;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int nt2 = w2 * 4 + j;
      const s16x4 y0 = lds_read_tr16(sy + mA * 512 + 32 * (nt2 ^ swA) + 8 * p);
      const s16x4 y1 = lds_read_tr16(sy + mB * 512 + 32 * (nt2 ^ swB) + 8 * p);
      const s16x8 yy = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
      yf[j] = __builtin_bit_cast(bf16x8, yy);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int nt1 = w1 * 8 + i;
      const s16x4 x0 = lds_read_tr16(sx + mA * 512 + 32 * (nt1 ^ swA) + 8 * p);
      const s16x4 x1 = lds_read_tr16(sx + mB * 512 + 32 * (nt1 ^ swB) + 8 * p);
      const s16x8 xx = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      xf[i] = __builtin_bit_cast(bf16x8, xx);
    }
  };
  auto bias_mfma = [&](const bf16x8* xf) __attribute__((always_inline)) {
    // static fragment indices (a runtime-indexed register array would go to scratch): select by w2
    const bf16x8 xa = w2 == 0 ? xf[0] : (w2 == 1 ? xf[2] : (w2 == 2 ? xf[4] : xf[6]));
    const bf16x8 xb = w2 == 0 ? xf[1] : (w2 == 1 ? xf[3] : (w2 == 2 ? xf[5] : xf[7]));
    accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_frag(), xa, accb[0], 0, 0, 0);
    accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_frag(), xb, accb[1], 0, 0, 0);
  };
  stage(0); stage(1); stage(2);
  bf16x8 xf[8], yf[4];
  if (w1 == 0) {
    for (int ks = 0; ks < nks; ks++) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // own pieces of slice ks (ks+1, ks+2 may fly)
      __builtin_amdgcn_s_barrier();                     // barrier 2ks: slice ks readable; everyone's reads of slice ks-1 are retired
      stage(ks + 3);                                    // into the slot of slice ks-1
      read_frags(ks, xf, yf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // barrier 2ks+1
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[j], xf[i], acc[i][j], 0, 0, 0);  // D[n2][n1]
      if (do_sum) bias_mfma(xf);
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the barrier group 1 executed in front
  } else {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // own pieces of slice 0
    __builtin_amdgcn_s_barrier();                       // barrier 0
    for (int ks = 0; ks < nks; ks++) {
      __builtin_amdgcn_s_barrier();                     // barrier 2ks+1: slice ks readable for this group (pieces waited before barrier 2ks)
      stage(ks + 3);                                    // slot of slice ks-1: group 0 retired its reads before barrier 2ks, this group before 2ks-1... 2ks
      read_frags(ks, xf, yf);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");  // own reads returned; own pieces of slice ks+1 landed (ks+2, ks+3 may fly)
      __builtin_amdgcn_s_barrier();                     // barrier 2ks+2
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[j], xf[i], acc[i][j], 0, 0, 0);
      if (do_sum) bias_mfma(xf);
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  if constexpr (SPLIT) if (sp.part >= 0) {   // half of the token range (gemm_tn_grouped, split launch): the fp32 tile + the column sums travel to / come from the other half (smbv_common.h)
    // the 128 accumulator registers go out / come in where they are (a copy of them would not fit the register file), the column sums as a second group
    // (row by row: the array is never addressed as a whole, which would put it into scratch memory)
    f32x4 sums[4] = {accb[0], accb[1], (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4* bias_base = sp.scratch + 32 * 512;
    f32x4* tp = sp.scratch + (size_t)w * (32 * 64) + l;   // piece 4 i + j of wave w at [(32 w + 4 i + j) * 64 + lane]
    __syncthreads();   // (both wave groups have left the ring: its first word carries the consumer's verdict)
    if (sp.part == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        sweep_store16_sc1<0>(tp + i * 256, acc[i][0]);
        sweep_store16_sc1<1024>(tp + i * 256, acc[i][1]);
        sweep_store16_sc1<2048>(tp + i * 256, acc[i][2]);
        sweep_store16_sc1<3072>(tp + i * 256, acc[i][3]);
      }
      sweep_store<4>(bias_base, sums, w, l);
      sweep_signal(sp, t);
      return;
    }
    if (!sweep_wait(sp, t, (int*)smem)) return;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      f32x4 a[4];
#pragma unroll
      for (int j = 0; j < 4; j++) a[j] = tp[i * 256 + j * 64];
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] += a[j];
      asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3])::"memory");
    }
    sweep_load<4>(bias_base, sums, w, l);
    sweep_release(sp, t);
    accb[0] = sums[0]; accb[1] = sums[1];
  }
  if (do_sum && l < 16) {
#pragma unroll
    for (int k = 0; k < 2; k++) tn_store_sum(ts, xsum_flat, a0 + w1 * 128 + (2 * w2 + k) * 16 + l, accb[k][0]);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int n1 = a0 + w1 * 128 + i * 16 + (l & 15);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n2 = b0 + w2 * 64 + j * 16 + 4 * (l >> 4);
      *(f32x4*)(o + (size_t)n1 * ldo + n2) = acc[i][j];
    }
  }
}

__global__ __launch_bounds__(512, 1) void gemm_tn_256_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ Y, int ldy,
                                                              int M, int N1, int N2, float* __restrict__ out, int ldo, int tiles_1,
                                                              int mchunk, size_t slab_stride, float* __restrict__ xsum) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 2 * 32 * 512];
  const int tile_1 = blockIdx.x % tiles_1, tile_2 = blockIdx.x / tiles_1;
  const int z = blockIdx.y;
  const int mbeg = z * mchunk, mend = min(M, mbeg + mchunk);
  const TnSum none = {nullptr, nullptr, nullptr, 0};
  tn256_tile(X, ldx, Y, ldy, N1, N2, out + (size_t)z * slab_stride, ldo, tile_1 * 256, tile_2 * 256, mbeg, mend, xsum != nullptr && tile_2 == 0,
             xsum ? xsum + (size_t)z * N1 : nullptr, none, smem);
}

// ---- grouped launch: several weight-gradient products (one transformer block's four) as ONE grid, every tile reduced over its product's WHOLE token
// range: no slabs, no finishing launch, the bias gradients (column sums of X) written straight to their tensors.  block.hip defers a block's products
// to the end of its backward and runs this launch on the side stream beside the NEXT block's backward.
template <int TILE, bool SPLIT = false>
__global__ __launch_bounds__(TILE == 256 ? 512 : 256, TILE == 256 ? 1 : 2) void gemm_tn_grouped_kernel(TnGroup g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TILE == 256 ? 4 * 2 * 32 * 512 : 2 * 2 * BK * 128 * 2];
  // split launch (g.tiles > 0: twice as many blocks as tiles): unit u < tiles reduces the FIRST half of tile u's token range and hands its fp32 tile to unit
  // tiles + u, which reduces the second half, adds and stores (the LAST grouped launch of a backward: nothing runs beside it, the optimizer waits for it).
  // Units are claimed from a ticket, not read off blockIdx: a consumer unit is then only ever held by a block that started after every producer
  // unit had been claimed by a RUNNING block - forward progress no longer rests on the order in which the dispatcher places workgroups.
  int unit = (int)blockIdx.x;
  SweepPart sp;
  if (SPLIT && g.tiles > 0) {
    int* bc = (int*)smem;
    if (threadIdx.x == 0) bc[0] = (int)(atomicAdd(g.ticket, 1u) - g.ticket0);
    __syncthreads();
    unit = bc[0];
    __syncthreads();   // (read by every wave before the ring overwrites it)
    if ((unsigned)unit >= 2u * (unsigned)g.tiles) {   // a counter another launch moved (two launches sharing a slot, a slot left dirty): no tile is run on a guess
      if (threadIdx.x == 0 && g.status) __hip_atomic_store(g.status, 0x5005u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    sp.part = unit >= g.tiles ? 1 : 0;
    unit -= sp.part * g.tiles;
    sp.scratch = g.scratch + (size_t)unit * (36 * 512);
    sp.flag = g.flags + unit;
    sp.want = g.want;
    sp.status = g.status;
  }
  int j = 0;
#pragma unroll
  for (int i = 1; i < TN_GROUP_MAX; i++)
    if (i < g.njobs && unit >= g.job[i].tile0) j = i;
  const TnJob& J = g.job[j];
  const int tile = unit - J.tile0;
  const int tiles_1 = J.N1 / TILE;
  const int tile_1 = tile % tiles_1, tile_2 = tile / tiles_1;
  const TnSum ts = {J.sum_dst[0], J.sum_dst[1], J.sum_dst[2], J.seg};
  const bool do_sum = J.seg > 0 && tile_2 == 0;
  const int mh = (J.M / 2) & ~31;   // (slices are 32 tokens)
  if constexpr (TILE == 256) tn256_tile<SPLIT>(J.X, J.ldx, J.Y, J.ldy, J.N1, J.N2, J.out, J.N2, tile_1 * 256, tile_2 * 256, sp.part == 1 ? mh : 0, sp.part == 0 ? mh : J.M,
                                               do_sum, nullptr, ts, smem, sp);
  else tn128_tile(J.X, J.ldx, J.Y, J.ldy, J.N1, J.N2, J.out, J.N2, tile_1 * 128, tile_2 * 128, 0, J.M, do_sum, nullptr, ts, smem);
}

// slabs for the 256 x 256 kernel: tiles x slabs as close to one full round of 256 blocks as the token count allows (>= 256 tokens a slab)
static int gemm_tn_splits_256(int M, int N1, int N2) {
  const int tiles = (N1 / 256) * (N2 / 256);
  int s = std::max(1, 256 / tiles);
  while (s > 1 && M / s < 256) s--;
  return s;
}
static bool gemm_tn_use_256(int M, int N1, int N2) {
  if ((N1 % 256) || (N2 % 256) || (g_variant & (32 | 32768))) return false;
  const int tiles = (N1 / 256) * (N2 / 256);
  const int s = gemm_tn_splits_256(M, N1, N2);
  // measured (tools/gemm_tn_ab.py): ahead of the 128 x 128 kernel when a slab still holds >= ~900 tokens (36-48 tiles: +3..26 %), behind it
  // when many short slabs are needed (9-27 tiles: -10..14 %)
  return ((g_variant & 65536) ? tiles >= 1 : tiles >= 32) && tiles <= 256 && tiles * s >= 192;
}

// One finishing launch per weight-gradient product: blocks [0, nb_slab) sum the token slabs (four in flight, slab order: reproducible), blocks
// [nb_slab, ...) sum the per-slab column sums of X into the bias gradient's up to three segments (round 2 ran these as two launches per product:
// 133 of the step's launches)
__global__ void reduce_tn_kernel(float* __restrict__ out, const float* __restrict__ slabs, int nslab, size_t n4, size_t slab_stride4, int nb_slab,
                                 const float* __restrict__ xpart, int N1, TnColsum dst) {
  if ((int)blockIdx.x >= nb_slab) {
    const int n = (blockIdx.x - nb_slab) * blockDim.x + threadIdx.x;
    if (n >= N1) return;
    float s = 0.f;
    for (int z = 0; z < nslab; z++) s += xpart[(size_t)z * N1 + n];  // slab order: reproducible
    float* d = dst.dst[n / dst.seg];
    if (d) d[n % dst.seg] = s;
    return;
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)nb_slab * blockDim.x) {
    const f32x4* p = (const f32x4*)slabs + i;
    f32x4 s = p[0];
    int z = 1;
    for (; z + 7 < nslab; z += 8) {   // eight slabs in flight: the pass is bound by load latency, not by bytes 
      const f32x4 a = p[z * slab_stride4], b = p[(z + 1) * slab_stride4], c = p[(z + 2) * slab_stride4], d = p[(z + 3) * slab_stride4];
      const f32x4 e = p[(z + 4) * slab_stride4], f = p[(z + 5) * slab_stride4], g = p[(z + 6) * slab_stride4], h = p[(z + 7) * slab_stride4];
      s += a; s += b; s += c; s += d; s += e; s += f; s += g; s += h;
    }
    for (; z + 3 < nslab; z += 4) {
      const f32x4 a = p[z * slab_stride4], b = p[(z + 1) * slab_stride4], c = p[(z + 2) * slab_stride4], d = p[(z + 3) * slab_stride4];
      s += a; s += b; s += c; s += d;
    }
    for (; z < nslab; z++) s += p[z * slab_stride4];
    ((f32x4*)out)[i] = s;
  }
}

// Token-range splits: blocks = tiles * s must fill the 512 block slots (2 per CU) of ONE round as well as possible - a grid of
// 576 blocks costs two rounds - while every block keeps a loop of at least ~16 K-steps (1024 tokens; 512 for tiny outputs).
int gemm_tn_splits(int M, int N1, int N2) {
  const int tiles = cdiv(N1, 128) * cdiv(N2, 128);
  int best = 1;
  double best_eff = 0.0;
  for (int pass = 0; pass < 2 && best_eff < 0.55; pass++) {
    const int min_rows = pass == 0 ? 1024 : 512;
    for (int s = 1; s <= 32; s++) {
      if (s > 1 && M / s < min_rows) break;
      const long blocks = (long)tiles * s;
      const long rounds = (blocks + 511) / 512;
      const double eff = (double)blocks / (double)(rounds * 512);
      if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
    }
  }
  return best;
}

size_t gemm_tn_workspace_bytes(int M, int N1, int N2) {
  const int s = std::max(gemm_tn_splits(M, N1, N2), (N1 % 256 == 0 && N2 % 256 == 0) ? gemm_tn_splits_256(M, N1, N2) : 1);
  // slabs (when the token range is split) + the per-slab column sums of X (bias gradient), 256-B aligned behind them
  return (s > 1 ? (((size_t)s * N1 * N2 * sizeof(float) + 255) & ~(size_t)255) : 0) + (size_t)s * N1 * sizeof(float);
}

static int tn_group_tile(const TnJob* jobs, int njobs) {  // 256, 128, or 0 = not groupable
  if (njobs <= 0 || njobs > TN_GROUP_MAX || (g_variant & 32)) return 0;
  bool all256 = !(g_variant & 32768), all128 = true;
  for (int i = 0; i < njobs; i++) {
    const TnJob& j = jobs[i];
    if (j.M <= 0 || j.N1 <= 0 || j.N2 <= 0 || (j.ldx & 7) || (j.ldy & 7) || !j.X || !j.Y || !j.out) return 0;
    if ((((uintptr_t)j.X | (uintptr_t)j.Y | (uintptr_t)j.out) & 15)) return 0;
    if ((size_t)j.M * j.ldx * 2 >= (1ull << 31) || (size_t)j.M * j.ldy * 2 >= (1ull << 31)) return 0;   // 32-bit LDS-DMA offsets
    if (j.seg < 0 || (j.seg > 0 && (j.N1 % j.seg || j.N1 / j.seg > 3))) return 0;
    all256 = all256 && j.N1 % 256 == 0 && j.N2 % 256 == 0;
    all128 = all128 && j.N1 % 128 == 0 && j.N2 % 128 == 0;
  }
  return all256 ? 256 : (all128 ? 128 : 0);
}
bool gemm_tn_grouped_ok(const TnJob* jobs, int njobs) { return tn_group_tile(jobs, njobs) != 0; }

// Device-side state of the split grouped launch: eight slots per device, a slot OWNED by one stream (the first eight streams that run a split launch; a ninth
// runs its launches whole), as the attention backward's pair launch keeps its queues.  `ticket` only ever counts up - a launch claims the run
// [ticket0, ticket0 + 2 tiles) the host reserved for it - and a flag is raised to the launch's sequence number, so neither a word left behind by a launch that
// faulted or gave up waiting (0x5004 / 0x5005) nor a launch of another stream is ever taken for this launch's.  smbv_debug_pair_state(0) restores it.
static constexpr int TN_SPLIT_TILES_MAX = 128;
struct TnSplitState { unsigned ticket; unsigned pad[63]; unsigned flag[TN_SPLIT_TILES_MAX]; };
__device__ TnSplitState g_tn_split[8];
struct TnSplitHost {
  hipStream_t streams[8];
  unsigned seq[8], base[8];
  TnSplitState* dev = nullptr;
  int n = 0;
};
static TnSplitHost g_tn_host[64];
static std::mutex g_tn_mu;

__global__ void tn_split_state_debug_kernel(int op, unsigned arg) {
  TnSplitState& s = g_tn_split[blockIdx.x];
  if (op == 1) { if (threadIdx.x == 0) atomicAdd(&s.ticket, 3u); return; }                                              // what a second launch sharing the slot would do
  if (op == 2) { for (int i = threadIdx.x; i < TN_SPLIT_TILES_MAX; i += blockDim.x) s.flag[i] = arg + 1u; return; }     // flags a launch of sequence number arg + 1 never saw consumed
  if (threadIdx.x == 0) s.ticket = 0;
  for (int i = threadIdx.x; i < TN_SPLIT_TILES_MAX; i += blockDim.x) s.flag[i] = 0;
}
int tn_split_state_debug(int op, unsigned arg, hipStream_t st) {
  hipLaunchKernelGGL(tn_split_state_debug_kernel, dim3(8), dim3(128), 0, st, op, arg);
  if (op == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      std::lock_guard<std::mutex> lk(g_tn_mu);
      g_tn_host[dev].n = 0;
    }
  }
  return check_launch("tn_split_state_debug");
}
// reserves the next launch of (current device, st): its slot's state, ticket run and sequence number; false = no slot (all eight belong to other streams)
struct TnSplitTake { TnSplitState* state; unsigned ticket0, want; int dev, slot; };
static bool tn_split_take(hipStream_t st, int units, TnSplitTake& t) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lk(g_tn_mu);
  TnSplitHost& h = g_tn_host[dev];
  if (!h.dev) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tn_split)) != hipSuccess) { (void)hipGetLastError(); return false; }
    h.dev = (TnSplitState*)p;
  }
  int slot = -1;
  for (int i = 0; i < h.n; i++) if (h.streams[i] == st) { slot = i; break; }
  if (slot < 0) {
    if (h.n >= 8) return false;
    slot = h.n++;
    h.streams[slot] = st; h.seq[slot] = 0; h.base[slot] = 0;
  }
  t.state = h.dev + slot; t.dev = dev; t.slot = slot;
  t.ticket0 = h.base[slot];
  h.base[slot] += (unsigned)units;
  if (++h.seq[slot] == 0) h.seq[slot] = 1;
  t.want = h.seq[slot];
  return true;
}
static void tn_split_untake(const TnSplitTake& t, int units) {   // the launch was not enqueued: the device counter never moved
  std::lock_guard<std::mutex> lk(g_tn_mu);
  TnSplitHost& h = g_tn_host[t.dev];
  if (t.slot < h.n && h.base[t.slot] == t.ticket0 + (unsigned)units) { h.base[t.slot] = t.ticket0; h.seq[t.slot] = t.want - 1u; }
}
size_t gemm_tn_grouped_split_bytes(int tiles) { return (size_t)tiles * 36 * 512 * sizeof(f32x4); }

int gemm_tn_grouped(const TnJob* jobs, int njobs, hipStream_t st, void* split_ws, size_t split_ws_bytes) {
  const int tile = tn_group_tile(jobs, njobs);
  if (!tile) { set_error("gemm_tn_grouped: products not groupable (1..%d jobs, widths multiples of 128, 16-byte aligned operands)", TN_GROUP_MAX); return SMBV_ERR_SHAPE; }
  TnGroup g;
  g.njobs = njobs;
  int tiles = 0;
  for (int i = 0; i < njobs; i++) {
    g.job[i] = jobs[i];
    g.job[i].tile0 = tiles;
    tiles += (jobs[i].N1 / tile) * (jobs[i].N2 / tile);
  }
  for (int i = njobs; i < TN_GROUP_MAX; i++) { g.job[i] = g.job[0]; g.job[i].tile0 = 0x7fffffff; }
  g.tiles = 0; g.scratch = nullptr; g.flags = nullptr; g.status = nullptr; g.ticket = nullptr; g.ticket0 = 0; g.want = 1u;
  // split launch: every tile's token range as two halves by two workgroups (256 x 256 tiles only; every product with at least 128 tokens per half).  The
  // launch's scheduling words live in a slot of the library's that belongs to `st` (above); a ninth stream of one device runs whole launches
  TnSplitTake take{};
  bool split = false;
  if (tile == 256 && split_ws && !((uintptr_t)split_ws & 255) && tiles <= TN_SPLIT_TILES_MAX && split_ws_bytes >= gemm_tn_grouped_split_bytes(tiles) && !(g_ab & 4)) {
    bool ok = true;
    for (int i = 0; i < njobs; i++) ok = ok && jobs[i].M >= 256;
    split = ok && tn_split_take(st, 2 * tiles, take);
    if (split) {
      g.tiles = tiles; g.scratch = (f32x4*)split_ws; g.flags = take.state->flag; g.status = device_status_word_dev();
      g.ticket = &take.state->ticket; g.ticket0 = take.ticket0; g.want = take.want;
    }
  }
  if (tile == 256 && g.tiles > 0) hipLaunchKernelGGL((gemm_tn_grouped_kernel<256, true>), dim3(2 * tiles), dim3(512), 0, st, g);
  else if (tile == 256) hipLaunchKernelGGL((gemm_tn_grouped_kernel<256, false>), dim3(tiles), dim3(512), 0, st, g);
  else hipLaunchKernelGGL((gemm_tn_grouped_kernel<128, false>), dim3(tiles), dim3(256), 0, st, g);
  const int rc = check_launch("gemm_tn_grouped");
  if (rc && split) tn_split_untake(take, 2 * tiles);
  return rc;
}

int gemm_tn(const bf16_t* X, int ldx, const bf16_t* Y, int ldy, int M, int N1, int N2, float* out, void* ws, size_t ws_bytes, hipStream_t st,
            const TnColsum* colsum) {
  if (M <= 0 || N1 <= 0 || N2 <= 0) { set_error("gemm_tn: empty problem"); return SMBV_ERR_SHAPE; }
  if ((N1 & 7) || (N2 & 7) || (ldx & 7) || (ldy & 7)) { set_error("gemm_tn: N1,N2,ldx,ldy must be multiples of 8 (N1=%d N2=%d ldx=%d ldy=%d)", N1, N2, ldx, ldy); return SMBV_ERR_SHAPE; }
  if (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)out) & 15) { set_error("gemm_tn: operands must be 16-byte aligned"); return SMBV_ERR_ALIGN; }
  if (colsum && (colsum->seg <= 0 || N1 % colsum->seg || N1 / colsum->seg > 3)) { set_error("gemm_tn: column-sum segments %d do not tile N1=%d", colsum->seg, N1); return SMBV_ERR_ARG; }
  const bool small32 = ((size_t)M * ldx * 2 < (1ull << 31)) && ((size_t)M * ldy * 2 < (1ull << 31));
  const bool big = gemm_tn_use_256(M, N1, N2) && small32;
  const int s = big ? gemm_tn_splits_256(M, N1, N2) : gemm_tn_splits(M, N1, N2);
  const size_t slab_bytes = s > 1 ? (((size_t)s * N1 * N2 * sizeof(float) + 255) & ~(size_t)255) : 0;
  const size_t need = slab_bytes + (colsum ? (size_t)s * N1 * sizeof(float) : 0);
  if (need && (!ws || ws_bytes < need || ((uintptr_t)ws & 15))) { set_error("gemm_tn: workspace too small (%zu < %zu)", ws_bytes, need); return SMBV_ERR_WORKSPACE; }
  float* slabs = s > 1 ? (float*)ws : out;
  float* xs = colsum ? (float*)((char*)ws + slab_bytes) : nullptr;
  const size_t stride = s > 1 ? (size_t)N1 * N2 : (size_t)0;
  if (big) {
    const int mchunk = cdiv(cdiv(M, s), 32) * 32;
    hipLaunchKernelGGL(gemm_tn_256_kernel, dim3((N1 / 256) * (N2 / 256), s), dim3(512), 0, st, X, ldx, Y, ldy, M, N1, N2, slabs, N2, N1 / 256, mchunk, stride, xs);
  } else {
    const int t1 = cdiv(N1, 128), t2 = cdiv(N2, 128);
    const bool dma = (N1 % 128 == 0) && (N2 % 128 == 0) && small32 && !(g_variant & 32);
    const int mchunk = cdiv(cdiv(M, s), 64) * 64;
    if (dma) hipLaunchKernelGGL(gemm_tn_dma_kernel, dim3(t1 * t2, s), dim3(256), 0, st, X, ldx, Y, ldy, M, N1, N2, slabs, N2, t1, mchunk, stride, xs);
    else hipLaunchKernelGGL(gemm_tn_kernel, dim3(t1 * t2, s), dim3(256), 0, st, X, ldx, Y, ldy, M, N1, N2, slabs, N2, t1, mchunk, stride, xs);
  }
  int rc = check_launch("gemm_tn");
  if (rc) return rc;
  if (s > 1 || colsum) {
    const size_t n4 = (size_t)N1 * N2 / 4;
    const int nb_slab = s > 1 ? (int)std::min<size_t>(2048, (n4 + 255) / 256) : 0, nb_x = colsum ? cdiv(N1, 256) : 0;
    hipLaunchKernelGGL(reduce_tn_kernel, dim3(nb_slab + nb_x), dim3(256), 0, st, out, (const float*)ws, s, n4, n4, nb_slab, (const float*)xs, N1,
                       colsum ? *colsum : TnColsum());
    rc = check_launch("reduce_tn");
  }
  return rc;
}

}  // namespace smbv
