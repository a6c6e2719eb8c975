// LayerNorm forward / backward over the last dimension (HBM-bound: one wave per row, wavefront-shuffle reductions,
// fp32 statistics).  Replaces the nn.LayerNorm call sites of src/models/videomae/modeling_videomae.py:402-403, 412,
// 423 (eps = layer_norm_eps) and :676/:721 (decoder.norm, torch-default eps 1e-5).
//
// forward : x fp32 [rows, D] -> y bf16 [rows, D] (the bf16 GEMM operand), mean/rstd fp32 [rows]
// backward: dy bf16 [rows, D], x, mean, rstd, gamma -> dx fp32 (+ optional residual-stream gradient added in),
//           a bf16 copy of dx for the next GEMM, and per-block partial dgamma/dbeta slabs that a second kernel sums
//           (deterministic; no float atomics).
#include "smbv_common.h"
#include "smbv_kernels.h"

namespace smbv {

static constexpr int LN_MAXV = 8;  // float4 per lane -> D <= 2048

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                                     float eps, long x_grp_in, long x_grp_out, long x_row_off, float* __restrict__ y32) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;
  // optional input row remap (decoder.norm reads the last N_mask rows of every batch item)
  size_t xr = row;
  if (x_grp_in > 0) xr = (size_t)(row / x_grp_in) * x_grp_out + (row % x_grp_in) + x_row_off;
  const float* xp = x + xr * D;
  const int nv = D >> 2;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int c = l + 64 * i;
    v[i] = (c < nv) ? ((const f32x4*)xp)[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int c = l + 64 * i;
    if (c < nv) {
      f32x4 d = v[i] - mu;
      q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
  }
  const float rs = rsqrtf(wave_sum(q) / D + eps);
  if (l == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
  bf16_t* yp = y + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int c = l + 64 * i;
    if (c < nv) {
      const f32x4 g = ((const f32x4*)gamma)[c], b = ((const f32x4*)beta)[c];
      const f32x4 o = (v[i] - mu) * rs * g + b;
      u32x2 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
      ((u32x2*)yp)[c] = pk;
      if (y32) ((f32x4*)(y32 + (size_t)row * D))[c] = o;
    }
  }
}

int layernorm_fwd(const float* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd, int rows, int D,
                  float eps, long grp_in, long grp_out, long row_off, hipStream_t st, float* y32) {
  if (rows <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * LN_MAXV) { set_error("layernorm_fwd: need 0 < D <= %d, D %% 4 == 0 (D=%d rows=%d)", 64 * 4 * LN_MAXV, D, rows); return SMBV_ERR_SHAPE; }
  if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15 || ((uintptr_t)y & 7)) { set_error("layernorm_fwd: alignment"); return SMBV_ERR_ALIGN; }
  const int nv = cdiv(D, 256);
  dim3 grid(cdiv(rows, 4)), blk(256);
#define LN_F(NV) hipLaunchKernelGGL(ln_fwd_kernel<NV>, grid, blk, 0, st, x, gamma, beta, y, mean, rstd, rows, D, eps, grp_in, grp_out, row_off, y32)
  if (nv <= 1) LN_F(1); else if (nv <= 2) LN_F(2); else if (nv <= 3) LN_F(3); else if (nv <= 4) LN_F(4); else LN_F(8);
#undef LN_F
  return check_launch("layernorm_fwd");
}

// backward. grid-stride over rows; each wave accumulates dgamma/dbeta partials for the columns its lanes own.  8 waves per block, up to 512
// blocks: 16 waves per CU keep enough independent rows in flight for the three operand streams (one row per wave at a time is latency-bound:
// at 4 waves per block the kernel ran at a third of the HBM rate).
static constexpr int LNB_WAVES = 8;
template <int NV>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16_t* __restrict__ dx_bf16,
                                                     float* __restrict__ part, int rows, int D, long grp_in, long grp_out,
                                                     long row_off) {
  __shared__ float red[LNB_WAVES * 2 * 256 * NV];  // [wave][dgamma|dbeta][256*NV columns]
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nv = D >> 2;
  f32x4 dg[NV], db[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    dg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    db[i] = dg[i];
    const int c = l + 64 * i;
    g[i] = (c < nv) ? ((const f32x4*)gamma)[c] : dg[i];
  }
  for (int row = blockIdx.x * LNB_WAVES + w; row < rows; row += gridDim.x * LNB_WAVES) {
    size_t xr = row;
    if (grp_in > 0) xr = (size_t)(row / grp_in) * grp_out + (row % grp_in) + row_off;
    const float mu = mean[row], rs = rstd[row];
    const float* xp = x + xr * D;
    const bf16_t* dyp = dy + (size_t)row * D;
    f32x4 xh[NV], gy[NV], dr[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {  // the residual gradient is fetched with the other operands, not after the two reductions
      const int c = l + 64 * i;
      dr[i] = (dres && c < nv) ? ((const f32x4*)(dres + xr * D))[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = l + 64 * i;
      if (c < nv) {
        const u32x2 pk = ((const u32x2*)dyp)[c];
        const f32x4 d = {bf_lo(pk[0]), bf_hi(pk[0]), bf_lo(pk[1]), bf_hi(pk[1])};
        xh[i] = (((const f32x4*)xp)[c] - mu) * rs;
        gy[i] = d * g[i];
        dg[i] += d * xh[i];
        db[i] += d;
        s1 += gy[i][0] + gy[i][1] + gy[i][2] + gy[i][3];
        s2 += gy[i][0] * xh[i][0] + gy[i][1] * xh[i][1] + gy[i][2] * xh[i][2] + gy[i][3] * xh[i][3];
      } else {
        xh[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gy[i] = xh[i];
      }
    }
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = l + 64 * i;
      if (c < nv) {
        f32x4 o = (gy[i] - s1 - xh[i] * s2) * rs;
        o += dr[i];
        if (dx) ((f32x4*)(dx + xr * D))[c] = o;
        if (dx_bf16) {
          u32x2 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
          ((u32x2*)(dx_bf16 + xr * D))[c] = pk;
        }
      }
    }
  }
  // block reduce of the 4 waves' partials through LDS, then one slab row per block: part[block][2][D]
  float* r = red;
  const int per_wave = 2 * 256 * NV;
#pragma unroll
  for (int i = 0; i < NV; i++)
#pragma unroll
    for (int e = 0; e < 4; e++) {
      r[w * per_wave + (i * 64 + l) * 4 + e] = dg[i][e];
      r[w * per_wave + 256 * NV + (i * 64 + l) * 4 + e] = db[i][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 64 * LNB_WAVES) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int ww = 0; ww < LNB_WAVES; ww++) {
      a += r[ww * per_wave + c];
      b += r[ww * per_wave + 256 * NV + c];
    }
    part[((size_t)blockIdx.x * 2 + 0) * D + c] = a;
    part[((size_t)blockIdx.x * 2 + 1) * D + c] = b;
  }
}

static int ln_bwd_blocks(int rows) { return std::max(1, std::min(512, cdiv(rows, LNB_WAVES))); }
size_t layernorm_bwd_workspace_bytes(int rows, int D) { return (size_t)ln_bwd_blocks(rows) * 2 * D * sizeof(float); }

int layernorm_bwd(const bf16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                  float* dx, bf16_t* dx_bf16, float* dgamma, float* dbeta, int rows, int D, long grp_in, long grp_out,
                  long row_off, void* ws, size_t ws_bytes, hipStream_t st, bool launch_reduce) {
  if (rows <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * LN_MAXV) { set_error("layernorm_bwd: bad shape rows=%d D=%d", rows, D); return SMBV_ERR_SHAPE; }
  if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)dres) & 15 || ((uintptr_t)dy & 7) || ((uintptr_t)dx_bf16 & 7)) { set_error("layernorm_bwd: alignment"); return SMBV_ERR_ALIGN; }
  const int nblk = ln_bwd_blocks(rows);
  if (!ws || ws_bytes < layernorm_bwd_workspace_bytes(rows, D)) { set_error("layernorm_bwd: workspace too small"); return SMBV_ERR_WORKSPACE; }
  float* part = (float*)ws;
  const int nv = cdiv(D, 256);
  dim3 grid(nblk), blk(64 * LNB_WAVES);
#define LN_B(NV) hipLaunchKernelGGL(ln_bwd_kernel<NV>, grid, blk, 0, st, dy, x, mean, rstd, gamma, dres, dx, dx_bf16, part, rows, D, grp_in, grp_out, row_off)
  if (nv <= 1) LN_B(1); else if (nv <= 2) LN_B(2); else if (nv <= 3) LN_B(3); else if (nv <= 4) LN_B(4); else LN_B(8);
#undef LN_B
  int rc = check_launch("layernorm_bwd");
  if (rc || !launch_reduce) return rc;
  return reduce_cols(part, nblk, 2 * D, dgamma, dbeta, D, st);
}

// the parameter-gradient half of layernorm_bwd(..., launch_reduce = false): sums the per-block partials `ws` holds into dgamma / dbeta.  Nothing on the
// input-gradient path reads them, so the block backward runs it on its side stream (block.hip) instead of between two kernels of the critical path.
int layernorm_bwd_reduce(const void* ws, int rows, int D, float* dgamma, float* dbeta, hipStream_t st) {
  return reduce_cols((const float*)ws, ln_bwd_blocks(rows), 2 * D, dgamma, dbeta, D, st);
}

}  // namespace smbv
