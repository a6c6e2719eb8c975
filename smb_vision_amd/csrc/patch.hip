// Brick (3-D patch) readers over the volume pixel_values fp32 [B, T, 1, H, W]  (HBM-bound).
//
//   gather_bricks : selected tokens' ts x ps x ps bricks -> bf16 rows [rows, K] in Conv3d-weight order
//                   k = (kt*ps + kh)*ps + kw  (src/models/videomae/modeling_videomae.py:172-177,190-191: Conv3d with
//                   kernel == stride is a GEMM over bricks; token order (T',H',W') row-major = flatten(2)).
//                   Only the visible bricks are read: the reference computes all N and discards the masked rows (:134-137).
//   brick_stats   : per-brick mean and 1/(sqrt(var_unbiased)+1e-6) of the masked bricks (:859-861, norm_pix_loss).
//   mse_fwd/bwd   : mean((logits - label)^2) over all B*N_mask*K elements (:896-897) with the label
//                   (x - mean) * rinv recomputed from the fp32 volume (per-patch order (ts,ps,ps,C) == k for C = 1), so the
//                   218 MB label tensor is never materialised.  Deterministic two-stage reduction.
// A brick row (fixed kt,kh) is ps contiguous floats (64 B at ps = 16): every thread moves one float4.
#include "smbv_common.h"
#include "smbv_kernels.h"

namespace smbv {

struct BrickGeom {
  int T, H, W, ts, ps, gh, gw, N, K;
};

__device__ __forceinline__ size_t brick_elem_off(const BrickGeom& g, int b, int tok, int k) {
  const int dq = tok / (g.gh * g.gw), hq = (tok / g.gw) % g.gh, wq = tok % g.gw;
  const int kw = k % g.ps, kh = (k / g.ps) % g.ps, kt = k / (g.ps * g.ps);
  return (((size_t)b * g.T + dq * g.ts + kt) * g.H + hq * g.ps + kh) * g.W + wq * g.ps + kw;
}

__global__ __launch_bounds__(256) void gather_bricks_kernel(const float* __restrict__ pix, const int* __restrict__ idx, int per_batch,
                                                            bf16_t* __restrict__ out, BrickGeom g) {
  const int row = blockIdx.x;
  const int b = row / per_batch;
  const int tok = idx[row];
  bf16_t* o = out + (size_t)row * g.K;
  for (int f = threadIdx.x; f < (g.K >> 2); f += 256) {
    const f32x4 v = *(const f32x4*)(pix + brick_elem_off(g, b, tok, 4 * f));
    u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    ((u32x2*)o)[f] = pk;
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// One pass over the brick (round 4): a thread keeps its float4s in registers between the mean and the variance sweep (K <= 4096 * RES / 4: the
// path's 16^3 bricks fit RES = 4; larger bricks re-read the tail from L2 as before).  Two-pass arithmetic, unchanged results.
template <int RES>
__global__ __launch_bounds__(256) void brick_stats_kernel(const float* __restrict__ pix, const int* __restrict__ idx, int per_batch,
                                                          float* __restrict__ mean, float* __restrict__ rinv, BrickGeom g) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int b = row / per_batch;
  const int tok = idx[row];
  const int nf = g.K >> 2;
  f32x4 keep[RES];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < RES; i++) {
    const int f = threadIdx.x + 256 * i;
    keep[i] = f < nf ? *(const f32x4*)(pix + brick_elem_off(g, b, tok, 4 * f)) : (f32x4){0.f, 0.f, 0.f, 0.f};
    s += (keep[i][0] + keep[i][1]) + (keep[i][2] + keep[i][3]);
  }
  for (int f = threadIdx.x + 256 * RES; f < nf; f += 256) {
    const f32x4 v = *(const f32x4*)(pix + brick_elem_off(g, b, tok, 4 * f));
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mu = block_sum_256(s, red) / g.K;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < RES; i++) {
    if (threadIdx.x + 256 * i < nf) {
      const f32x4 v = keep[i] - mu;
      q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
  }
  for (int f = threadIdx.x + 256 * RES; f < nf; f += 256) {
    const f32x4 v = *(const f32x4*)(pix + brick_elem_off(g, b, tok, 4 * f)) - mu;
    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  const float var = block_sum_256(q, red) / (g.K - 1);  // unbiased
  if (threadIdx.x == 0) {
    mean[row] = mu;
    rinv[row] = 1.0f / (sqrtf(var) + 1e-6f);
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pix, const int* __restrict__ idx, int per_batch,
                                                  const float* __restrict__ mean, const float* __restrict__ rinv,
                                                  const bf16_t* __restrict__ logits, float* __restrict__ row_partial,
                                                  bf16_t* __restrict__ dlogits, const float* __restrict__ gscale, float coef,
                                                  BrickGeom g) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int b = row / per_batch;
  const int tok = idx[row];
  const float mu = mean[row], ri = rinv[row];
  const bf16_t* lp = logits + (size_t)row * g.K;
  float s = 0.f;
  float gs = coef;
  if (BWD && gscale) gs *= *gscale;
  for (int f = threadIdx.x; f < (g.K >> 2); f += 256) {
    const f32x4 v = (*(const f32x4*)(pix + brick_elem_off(g, b, tok, 4 * f)) - mu) * ri;
    const u32x2 pk = ((const u32x2*)lp)[f];
    const f32x4 d = (f32x4){bf_lo(pk[0]), bf_hi(pk[0]), bf_lo(pk[1]), bf_hi(pk[1])} - v;
    if (BWD) {
      const f32x4 o = d * gs;
      u32x2 po = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
      ((u32x2*)(dlogits + (size_t)row * g.K))[f] = po;
    } else {
      s += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  if (!BWD) {
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) row_partial[row] = tot;
  }
}

__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ part, int n, float coef, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0) *out = tot * coef;
}

static int make_geom(BrickGeom& g, int T, int H, int W, int ts, int ps, const char* who) {
  if (T <= 0 || H <= 0 || W <= 0 || ts <= 0 || ps <= 0 || T % ts || H % ps || W % ps || (ps & 3)) {
    set_error("%s: volume %dx%dx%d not divisible by tubelet %d / patch %d (patch must be a multiple of 4)", who, T, H, W, ts, ps);
    return SMBV_ERR_SHAPE;
  }
  g.T = T; g.H = H; g.W = W; g.ts = ts; g.ps = ps; g.gh = H / ps; g.gw = W / ps;
  g.N = (T / ts) * g.gh * g.gw;
  g.K = ts * ps * ps;
  return SMBV_OK;
}

int gather_bricks(const float* pix, const int* idx, int rows, int per_batch, bf16_t* out, int T, int H, int W, int ts, int ps, hipStream_t st) {
  BrickGeom g;
  int rc = make_geom(g, T, H, W, ts, ps, "gather_bricks");
  if (rc) return rc;
  if (rows <= 0 || per_batch <= 0 || rows % per_batch) { set_error("gather_bricks: rows=%d per_batch=%d", rows, per_batch); return SMBV_ERR_SHAPE; }
  if (((uintptr_t)pix & 15) || ((uintptr_t)out & 7)) { set_error("gather_bricks: alignment"); return SMBV_ERR_ALIGN; }
  hipLaunchKernelGGL(gather_bricks_kernel, dim3(rows), dim3(256), 0, st, pix, idx, per_batch, out, g);
  return check_launch("gather_bricks");
}

int brick_stats(const float* pix, const int* idx, int rows, int per_batch, float* mean, float* rinv, int T, int H, int W, int ts, int ps, hipStream_t st) {
  BrickGeom g;
  int rc = make_geom(g, T, H, W, ts, ps, "brick_stats");
  if (rc) return rc;
  if (rows <= 0 || per_batch <= 0 || rows % per_batch) { set_error("brick_stats: rows=%d per_batch=%d", rows, per_batch); return SMBV_ERR_SHAPE; }
  hipLaunchKernelGGL(brick_stats_kernel<4>, dim3(rows), dim3(256), 0, st, pix, idx, per_batch, mean, rinv, g);
  return check_launch("brick_stats");
}

int mse_fwd(const float* pix, const int* idx, int rows, int per_batch, const float* mean, const float* rinv, const bf16_t* logits,
            float* row_partial, float* loss, int T, int H, int W, int ts, int ps, hipStream_t st) {
  BrickGeom g;
  int rc = make_geom(g, T, H, W, ts, ps, "mse_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(mse_kernel<false>, dim3(rows), dim3(256), 0, st, pix, idx, per_batch, mean, rinv, logits, row_partial, (bf16_t*)nullptr,
                     (const float*)nullptr, 0.f, g);
  rc = check_launch("mse_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, row_partial, rows, 1.0f / ((float)rows * (float)g.K), loss);
  return check_launch("mse_reduce");
}

int mse_bwd(const float* pix, const int* idx, int rows, int per_batch, const float* mean, const float* rinv, const bf16_t* logits,
            bf16_t* dlogits, const float* gscale, int T, int H, int W, int ts, int ps, hipStream_t st) {
  BrickGeom g;
  int rc = make_geom(g, T, H, W, ts, ps, "mse_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(mse_kernel<true>, dim3(rows), dim3(256), 0, st, pix, idx, per_batch, mean, rinv, logits, (float*)nullptr, dlogits, gscale,
                     2.0f / ((float)rows * (float)g.K), g);
  return check_launch("mse_bwd");
}

}  // namespace smbv
